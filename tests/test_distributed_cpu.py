"""The N > 1 path on CPU: world_size 2, gloo.  One broadcast of the constants, frame sharding, MAX timing."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from pysilent_amd import distributed as D
    from pysilent_amd.pipeline import default_constants
    rank, world, local = D.init(backend="gloo")
    assert dist.get_world_size() == 2 and world == 2
    if rank != 0:
        # prove the receivers take rank 0's floats: poison the local generator
        import pysilent_amd.pipeline as P
        real = P.default_constants
        D.default_constants = lambda mode, n=4: {k: v * 0 - 1 for k, v in real(mode, n).items()}
    consts = D.broadcast_constants("gray", 4)
    want = default_constants("gray", 4)
    ok = all(np.array_equal(consts[k], want[k]) for k in want)
    rgb = D.broadcast_constants("rgb")
    ok = ok and all(np.array_equal(rgb[k], default_constants("rgb")[k]) for k in rgb)
    frames = D.shard_frame_indices(10, rank, world)
    slow = D.max_over_ranks(1.0 + rank)
    D.barrier()
    print(json.dumps({"rank": rank, "ok": bool(ok), "frames": frames, "slow": slow}))
    dist.destroy_process_group()
""") % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gloo_broadcast_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    import json
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert all(o["ok"] for o in outs)
    assert outs[0]["frames"] == [0, 2, 4, 6, 8] and outs[1]["frames"] == [1, 3, 5, 7, 9]
    assert outs[0]["slow"] == 2.0 and outs[1]["slow"] == 2.0


def test_single_process_world_is_trivial():
    from pysilent_amd import distributed as D
    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        assert D.world() == (0, 1, 0)
        c = D.broadcast_constants("gray", 4)
        assert c["end"].shape == (3, 3, 1, 4) and c["cs"].dtype == np.float32
        assert D.max_over_ranks(3.5) == 3.5
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v


def test_bench_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun: the parent starts the ranks itself (before any GPU call), relays
    rank 0's JSON line and exits 0.  SILENT_BENCH_DRY=1 keeps the ranks off the GPU: rendezvous, the one broadcast of
    the constants, frame sharding, barrier and the MAX over ranks all run for real over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SILENT_BENCH_DRY="1", SILENT_DIST_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    assert p.stdout.strip() == lines[0]            # stdout holds the JSON line and nothing else (gloo / RCCL banners go to stderr)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    assert out["slowest_rank_time"] == 2.0           # MAX over ranks of (1 + rank)
    # the contract's `value`: the SUM over ranks of the frames they processed / the MAX over ranks of their times
    h, w = out["frame_hw"]
    frames_all_ranks = sum(out["frames_per_rank_per_step"] * out["steps"] for _ in out["dist"]["ranks"])
    slowest_s = max(r["ms_per_step"] for r in out["dist"]["ranks"])          # (dry run: "ms_per_step" = 1 + rank seconds)
    assert out["value"] == frames_all_ranks * h * w / slowest_s / 1e6
    assert out["frames_of_rank0"] == [0, 2, 4, 6]    # frame i -> rank i mod N
    assert out["constants"] == ["cs", "end"]
    # the scaling record proves itself: backend, world size, and per rank the device it ran on and its own time
    d = out["dist"]
    assert d["backend"].startswith("gloo") and d["world_size"] == 2 and d["distinct_devices"] == 2
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    assert [r["ms_per_step"] for r in d["ranks"]] == [1.0, 2.0]
    assert len({r["pci_bus_id"] for r in d["ranks"]}) == 2 and len({r["pid"] for r in d["ranks"]}) == 2
    for r in d["ranks"]:
        assert set(r) >= {"rank", "local_rank", "device_name", "pci_bus_id", "ms_per_step", "host", "pid"}


def test_bench_launcher_eight_ranks_dry_run():
    """The driver's N = 8 case rehearsed on the CPU: eight fresh rank processes, one gloo rendezvous on 127.0.0.1, the broadcast of
    the constants, frame i -> rank i mod 8, MAX over ranks -- and every rank's record says what its tuners would run with: on N > 1
    the overlap policy is "off" (one deterministic one-stream step per rank, no eight tuners timing candidates beside each other)
    unless SILENT_OVERLAP says otherwise."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SILENT_OVERLAP")}
    env.update(SILENT_BENCH_DRY="1", SILENT_DIST_BACKEND="gloo", SILENT_BENCH_LAUNCH_TIMEOUT="240", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    d = out["dist"]
    assert out["n_gpus"] == 8 and d["world_size"] == 8 and d["distinct_devices"] == 8
    assert [r["rank"] for r in d["ranks"]] == list(range(8)) and len({r["pid"] for r in d["ranks"]}) == 8
    assert out["slowest_rank_time"] == 8.0 and out["frames_of_rank0"] == [0, 8, 16, 24]
    h, w = out["frame_hw"]
    assert out["value"] == 8 * out["frames_per_rank_per_step"] * out["steps"] * h * w / 8.0 / 1e6
    for r in d["ranks"]:
        assert r["overlap_policy"] == "False" and r["streams"] == "one" and "settle_steps_run" in r and "placement_chosen_ms" in r


def test_overlap_policy_is_one_decision():
    """SILENT_OVERLAP=off|on|auto; unset: measured ("auto") on one GPU, off on N > 1."""
    import bench
    old = os.environ.pop("SILENT_OVERLAP", None)
    try:
        assert bench.overlap_policy(1) == "auto" and bench.overlap_policy(2) is False and bench.overlap_policy(8) is False
        for v, want in (("off", False), ("on", "auto"), ("auto", "auto"), ("0", False)):
            os.environ["SILENT_OVERLAP"] = v
            assert bench.overlap_policy(1) == want and bench.overlap_policy(8) == want
    finally:
        os.environ.pop("SILENT_OVERLAP", None)
        if old is not None:
            os.environ["SILENT_OVERLAP"] = old


def test_bench_refuses_two_ranks_on_one_device():
    """Two ranks that report the same PCI bus id: exit non-zero, no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SILENT_BENCH_DRY="1", SILENT_DIST_BACKEND="gloo", SILENT_BENCH_DRY_SAME_BUS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "share a GPU" in p.stderr


def test_bench_launcher_dumps_the_failing_ranks_log(tmp_path):
    """Rank 1 dies: the launcher exits with its code and prints the tail of rank 1's own log file."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SILENT_BENCH_DRY="1", SILENT_DIST_BACKEND="gloo", SILENT_BENCH_DRY_FAIL_RANK="1",
               SILENT_BENCH_LOG_DIR=str(tmp_path), SILENT_BENCH_LAUNCH_TIMEOUT="120")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 7
    assert "rank 1 exited with code 7" in p.stderr and "fails on purpose" in p.stderr
    assert (tmp_path / "rank1.log").exists() and (tmp_path / "rank0.log").exists()


def test_bench_launcher_reports_a_failed_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SILENT_BENCH_DRY="1", SILENT_DIST_BACKEND="no-such-backend", SILENT_BENCH_LAUNCH_TIMEOUT="120")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert "rank" in p.stderr


def test_bench_launcher_stops_a_hung_rank(tmp_path):
    """Rank 1 never reaches the rendezvous' first barrier: after SILENT_BENCH_LAUNCH_TIMEOUT the launcher exits 124, prints the
    tails of the ranks still running, and leaves no process behind (rank 0 blocks in the barrier and is stopped too)."""
    import re
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SILENT_BENCH_DRY="1", SILENT_DIST_BACKEND="gloo", SILENT_BENCH_DRY_HANG_RANK="1",
               SILENT_BENCH_LOG_DIR=str(tmp_path), SILENT_BENCH_LAUNCH_TIMEOUT="20")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 124, p.stderr[-2000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "still running after" in p.stderr and "hangs on purpose" in p.stderr
    pid = int(re.search(r"rank 1 \(pid (\d+)\) hangs", p.stderr).group(1))
    for _ in range(50):                       # the launcher has waited for the processes it terminated
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("rank 1 (pid %d) is still alive after the launcher returned" % pid)


def test_bench_launcher_survives_a_chatty_rank0(tmp_path):
    """A library that prints far more than a pipe buffer to file descriptor 1 of rank 0 BEFORE bench.py has redirected it must
    not block the launch: rank 0's stdout is a file, and the JSON line is still the last line relayed."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    site = tmp_path / "site"
    site.mkdir()
    (site / "sitecustomize.py").write_text("import os\nif os.environ.get('RANK') == '0':\n    os.write(1, b'x' * 300000 + b'\\n')\n")
    env.update(SILENT_BENCH_DRY="1", SILENT_DIST_BACKEND="gloo", SILENT_BENCH_LOG_DIR=str(tmp_path),
               PYTHONPATH=str(site) + os.pathsep + env.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1]
    assert json.loads(last)["dry_run"] is True
