#!/usr/bin/env python3
"""bench.py -- Mpx/s of the full pyramid line-end pass @1080p (BASELINE.json metric), one rank per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B] [--workload config2|config3|config5]

A "step" is one pass of the hot path over one batch of B device-resident synthetic frames per rank:
zoom pyramid -> center-surround -> ReLU -> oriented line-end bank -> ReLU -> clip (config 2/5) or the
reference's RGB chain + keypoints (config 3).  Frames shard over ranks (frame i -> rank i mod N, no
per-frame collective); the constant kernels are generated on rank 0 and broadcast once over RCCL.

Launching: under torchrun (RANK / WORLD_SIZE / LOCAL_RANK in the environment) this process is one rank.
Without them, ``--gpus N`` with N > 1 makes THIS process a launcher: it starts N fresh rank processes (one
per GPU, rendezvous on 127.0.0.1) BEFORE touching the GPU itself, relays rank 0's JSON line and exits
non-zero if any rank failed.  It never re-execs a process that has initialised the GPU.

Rank 0 prints ONE JSON line.  ``value`` = input-frame megapixels per second over all ranks, inputs
resident in HBM when the timed region starts; the timed region holds nothing but the K steps (no event
records, no host reads).  AFTER it, and outside ``value``: ``roofline`` prices the dominant kernel against
HBM from HIP events recorded around its launches in a loop of their own; ``other_workloads`` carries short
runs of configs 3 and 5; ``cpu_baseline`` times the CPU oracle on the host cores (N = 1 only).
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC: without this RCCL fails with `hipIpcGetMemHandle: invalid argument`
# (already exported on the GPU boxes; set here too so that a rank started under plain torchrun has it before HIP loads)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: 1080p grayscale, 5-level pyramid, center-surround + 4-orientation line-end
    "config2": dict(hw=(1080, 1920), mode="gray", n_levels=5, n_orient=4, frames=64,
                    name="1080p gray, 5-level pyramid (scale 2), CS + 4-orientation line-end"),
    # configs[2]: 1080p RGB, 6-level pyramid, normalize + peak extraction
    "config3": dict(hw=(1080, 1920), mode="rgb", n_levels=6, n_orient=3, frames=32,
                    name="1080p RGB, 6-level pyramid (scale 2), rgc>rgby>stripe>regulate>end>pad>value, top 10 %, NMS, keypoints"),
    # the reference's OWN pyramid layout (LineEndDisplayer defaults, pyramid_displayer.py:22: output_size (288, 192), zoom ratio
    # e ** .5; util/zoom/from_image.py:45-64: nested centre crops resampled to one fixed size) on batched 1080p RGB frames
    "reference_layout": dict(hw=(1080, 1920), mode="rgb", n_levels=4, n_orient=3, frames=32, center=(288, 192), scale=math.e ** .5,
                             name="1080p RGB, the reference's layout: 4 levels of 288x192 (centre crops, zoom e^-s/2), chain + keypoints"),
    # the same layout on one channel (PyramidDisplayer(output_colors=1), pyramid_displayer.py:22) with the gray chain of config 2
    "reference_layout_gray": dict(hw=(1080, 1920), mode="gray", n_levels=4, n_orient=4, frames=64, center=(288, 192), scale=math.e ** .5,
                                  name="1080p gray, the reference's layout: 4 levels of 288x192 (centre crops, zoom e^-s/2), CS + 4-orientation line-end"),
    # configs[4]: 4K, 8-level pyramid, 8-orientation bank
    "config5": dict(hw=(2160, 3840), mode="gray", n_levels=8, n_orient=8, frames=16,
                    name="4K gray, 8-level pyramid (scale 2), CS + 8-orientation line-end"),
}


def _metric_name():
    """BASELINE.json's own wording of the metric (the file travels with the repo); a literal copy as fallback."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except (OSError, ValueError, KeyError):
        return "Mpx/s full pyramid line-end pass @1080p, 1/2/4/8 GPU; % HBM roofline"


METRIC = _metric_name()
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
COPY_PEAK_GBS = 6290.0  # same guide: measured float4 copy


# ------------------------------------------------------------------------------------------ launcher (no GPU here)

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _tail(path, n=3000):
    try:
        with open(path, "rb") as f:
            f.seek(0, 2)
            size = f.tell()
            f.seek(max(0, size - n))
            return f.read().decode("utf-8", "replace")
    except OSError:
        return ""


def launch_ranks(n, argv, timeout_s=None):
    """Start n rank processes of this script (RANK = LOCAL_RANK = 0..n-1), relay rank 0's stdout, return the exit
    code (0 only if every rank exited 0).  The parent has not imported torch and makes no HIP call.  Rank r >= 1 writes
    stdout + stderr to <log dir>/rank<r>.log (SILENT_BENCH_LOG_DIR, default a fresh temporary directory); when a rank
    fails, the tail of ITS log is printed to stderr."""
    import tempfile
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    timeout_s = timeout_s or float(os.environ.get("SILENT_BENCH_LAUNCH_TIMEOUT", "1500"))
    logdir = os.environ.get("SILENT_BENCH_LOG_DIR") or tempfile.mkdtemp(prefix="silent_bench_")
    os.makedirs(logdir, exist_ok=True)
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        path = os.path.join(logdir, "rank%d.log" % r)
        logs.append(path)
        log = open(path, "wb")
        # rank 0: stdout is the JSON line (relayed), stderr goes to its log; ranks >= 1: both to the log
        # (rank 0's stdout goes to a FILE, not a pipe: nothing a library prints to fd 1 can fill a pipe buffer and block the
        # rank while this loop only polls exit codes)
        out0 = open(os.path.join(logdir, "rank0.out"), "wb") if r == 0 else None
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else log, stderr=log))
        log.close()
        if out0:
            out0.close()
    deadline = time.time() + timeout_s
    rc, failed = 0, None
    pending = set(range(n))
    while pending and time.time() < deadline and failed is None:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    failed, rc = r, code
        time.sleep(0.05)
    if pending:  # a rank failed or the deadline passed: stop exactly the processes started here
        for r in pending:
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        if failed is None:
            rc = 124
            print("bench.py launcher: ranks %s still running after %.0f s (logs: %s)" % (sorted(pending), timeout_s, logdir),
                  file=sys.stderr)
            for r in sorted(pending):
                print("---- rank %d, tail of %s ----\n%s" % (r, logs[r], _tail(logs[r])), file=sys.stderr)
    try:
        with open(os.path.join(logdir, "rank0.out"), "r", errors="replace") as f:
            out = f.read()
    except OSError:
        out = ""
    if failed is not None:
        print("bench.py launcher: rank %d exited with code %d; tail of %s:\n%s" % (failed, rc, logs[failed], _tail(logs[failed])),
              file=sys.stderr)
    else:
        err0 = _tail(logs[0])
        if err0.strip():
            sys.stderr.write(err0)
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc if rc else 0


# ------------------------------------------------------------------------------------------ measured traffic (PMC)

def csrc_revision():
    """sha256 over the kernel sources: the PMC summaries under profiles/ carry the revision they were taken at."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pysilent_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_per_launch(kernel_substr, frames_in_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/<tag>/pmc_hbm_bytes.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this bench), corrected as
    MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2; WRITE_SIZE exact;
    both in KiB.  Only a summary stamped with the CURRENT revision of csrc/ counts: after any kernel change the
    figure is null until scripts/profile_bench.sh + summarize_profile.py have been run again."""
    import glob
    rev = csrc_revision()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_hbm_bytes.json")), reverse=True):
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        if prof.get("_csrc_revision") != rev:
            continue
        for name, c in prof.items():
            if isinstance(c, dict) and kernel_substr in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                per = (2.0 * c["FETCH_SIZE"]["mean_KiB_per_dispatch"] + c["WRITE_SIZE"]["mean_KiB_per_dispatch"]) * 1024.0
                pmc_traffic_per_launch.source = "%s@%s (stored rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench at this csrc revision; not re-measured in this run)" % (
                    os.path.relpath(path, ROOT), rev)
                return int(per * frames_in_launch / c.get("frames_per_dispatch", 64))
    pmc_traffic_per_launch.source = None
    return None


pmc_traffic_per_launch.source = None


# ------------------------------------------------------------------------------------------ CPU baselines

def _native_oracle():
    """the C port is built -O3 -march=native ON the box that times it (`make -C oracle native`; the
    library that travels with the snapshot is -march=x86-64-v3).  Falls back to the portable build when gcc is missing."""
    path = os.path.join(ROOT, "oracle", "libsilent_oracle_native.so")
    try:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "-B", "native"], check=True, timeout=120,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.environ["SILENT_ORACLE_SO"] = path
        return "-O3 -march=native -fopenmp (built on this host)"
    except (OSError, subprocess.SubprocessError):
        return "-O3 -march=x86-64-v3 -fopenmp (portable build; `make native` failed here)"


def cpu_baseline(wl, consts, budget_s=12.0):
    """The CPU oracle timed on this box's host cores, on a bounded sample of the same workload; the oracle is the
    thing timed here, never part of the GPU path.
      B2 (the reported ``value``, kind "port"): oracle/silent_oracle.c, one frame per OpenMP thread.
      B1 ("numpy_scipy"): oracle/silent_oracle.py, the closest analogue of the reference's host path -- the same
          scipy.ndimage.zoom(order=5, prefilter=False) call for the pyramid, NumPy for the stencils, one process.
    gray: pyramid + CS + K-orientation line-end (configs 2 / 5); rgb: pyramid + the reference chain + top 10 % + NMS + value +
    per-region keypoint indices (config 3; the op list of recognition_testing.py:69-90)."""
    flags = _native_oracle()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle as co
    import silent_oracle as so
    from pysilent_amd.distributed import synthetic_frame
    h, w = wl["hw"]
    if "center" in wl:
        return None
    rgb = wl["mode"] == "rgb"
    c = 3 if rgb else 1
    extents = so.classic_extents(h, w, 2.0, wl["n_levels"])
    threads = co.num_threads()
    batch = max(threads, 1)
    frames = np.empty((batch, h, w, c), np.float32)
    for i in range(batch):
        frames[i] = synthetic_frame(i % 16, h, w, c)
    if rgb:
        ks = {k: consts[k] for k in ("rgc", "rgby", "stripe", "blur", "end")}
        run = lambda fr: co.rgb_pass_frames(fr, extents, ks, "ieee", 0.1)
        what = "pyramid + rgc>rgby>stripe>regulate>end>pad>value + top 10 % + NMS + keypoint indices"
    else:
        run = lambda fr: co.gray_pass_frames(fr, extents, consts["cs"], consts["end"])
        what = "pyramid + CS + %d-orientation line-end" % wl["n_orient"]
    run(frames[:min(batch, 8)])      # warm: page faults, OpenMP team
    n, total = 0, 0.0
    while total < budget_s and n < 64 * batch:
        t = time.perf_counter()
        run(frames)
        total += time.perf_counter() - t
        n += batch
    out = {"value": round(n * h * w / total / 1e6, 3), "unit": "Mpx/s", "cores": threads, "kind": "port",
           "sample": "%d synthetic %dx%dx%d frames, whole pass (%s), oracle/silent_oracle.c %s, one frame per thread, "
                     "float64 accumulation, %.1f s wall" % (n, w, h, c, what, flags, total),
           "host_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    # B1: a handful of frames through the NumPy / SciPy oracle
    n1, t1 = 0, 0.0
    while t1 < 8.0 and n1 < (2 if rgb else 8):
        frame = synthetic_frame(n1, h, w, c)
        t = time.perf_counter()
        pyr = so.classic_pyramid(frame, 2.0, wl["n_levels"])
        if rgb:
            for lev in pyr:
                line = so.rgb_line_end_chain(lev, ks)["padded"]
                top = so.top_value_points(line, 0.1)
                pv = so.value_from_color(so.nms3x3(top, "product"))
                so.max_value_indices_region(None, (1, max(lev.shape[1] // 2, 1), max(lev.shape[2] // 2, 1), 3), pv)
        else:
            so.gray_line_end_pass(pyr, consts["cs"], consts["end"])
        t1 += time.perf_counter() - t
        n1 += 1
    out["numpy_scipy"] = {"value": round(n1 * h * w / t1 / 1e6, 3), "unit": "Mpx/s", "kind": "reference-like",
                          "threads": "1 process; scipy.ndimage.zoom is single-threaded, NumPy stencils use its own BLAS-free loops",
                          "sample": "%d frames, oracle/silent_oracle.py (scipy.ndimage.zoom order=5 prefilter=False + NumPy), %.1f s"
                                    % (n1, t1)}
    return out


# ------------------------------------------------------------------------------------------ one rank

def overlap_policy(world):
    """Whether consecutive steps may overlap on two streams: SILENT_OVERLAP=off|on|auto.  Default: "auto" on one GPU (the pipeline
    measures stream pairs against its one-stream step and keeps what wins), "off" on N > 1 -- every rank then runs the same,
    deterministic one-stream step, and N tuners do not time their candidates beside each other on a shared host.  "on" = "auto"
    (a pair is never kept unmeasured)."""
    v = os.environ.get("SILENT_OVERLAP", "").strip().lower()
    if v in ("off", "0", "false", "no"):
        return False
    if v in ("on", "auto", "1", "true", "yes"):
        return "auto"
    return "auto" if world == 1 else False


def placement_policy():
    """SILENT_PLACEMENT=off: keep the first allocation of the maps; default: the product default, LineEndPipeline.tune_placement."""
    return os.environ.get("SILENT_PLACEMENT", "").strip().lower() not in ("off", "0", "false", "no")


def make_pipeline(wl, B, local, consts, **over):
    """The pipeline of a workload, UNTUNED (one stream, first allocation): tune_pipeline() does the measuring once the caller's
    frames are resident."""
    from pysilent_amd.pipeline import LineEndPipeline
    h, w = wl["hw"]
    kw = {}
    if wl["mode"] == "rgb":
        # config 3 returns line_end + keypoints (+ orient, optional in SURVEY.md section 8d): the value map and the selection's
        # peak-value map are intermediates the fused step never writes (silent_rgb_keypoints: extrema in the chain kernel,
        # sparse keypoint tail)
        kw = {"selection": True, "value_map": False, "peak_value_map": False}
    if "center" in wl:
        kw.update(center_dimensions=wl["center"], scale=wl["scale"])
    kw.setdefault("placement", None)      # (UNTUNED: tune_pipeline() runs the product's placement="auto" tuner explicitly, outside the timed region)
    kw.update(over)
    # keypoint capacity = every pyramid pixel of a frame (the LineEndPipeline default): a window without a positive peak makes
    # every pixel mapped to it a keypoint (top_value_points.py:32-45), noise frames produce ~10^5 .. 10^6 rows, and a smaller
    # cap would drop rows inside the timed region (round 3's 1 << 16 did)
    return LineEndPipeline((h, w), mode=wl["mode"], n_levels=wl["n_levels"], n_orient=wl["n_orient"], batch=B,
                           device=local, constants=consts, **kw)


def tune_pipeline(pipe, frames, overlap, placement=True):
    """Outside every timed region, once per pipeline: (1) placement -- what LineEndPipeline(placement="auto"), the product default,
    does on its first batch: the big map stays, the small maps are drawn again a few times behind spacers and the fastest
    relation is kept (the step's time depends on where the maps lie RELATIVE to each other in physical memory, by up to 25 %:
    LineEndPipeline.tune_placement, profiles/r06/placement.md; <= 1 s, the first draw's time is reported beside the chosen one); (2) overlap="auto" -- candidate stream pairs against the one-stream step, consecutive batches
    overlap on two streams only where that measurably pays (gray: the stream kernel of batch n + 1 beside the filter kernel of
    batch n; rgb: the pyramid beside chain + tail).  Both leave their record on the pipeline; results never depend on either."""
    if placement:
        pipe.tune_placement(frames)
    if overlap:
        pipe.tune_overlap(frames)


def make_frames(torch, D, wl, B, rank, world, dev):
    h, w = wl["hw"]
    c = 1 if wl["mode"] == "gray" else 3
    frames = torch.empty((B, h, w, c), dtype=torch.float32, device=dev)
    for j, gi in enumerate(D.shard_frame_indices(B * world, rank, world)):
        frames[j] = torch.from_numpy(D.synthetic_frame(gi, h, w, c)).to(dev)
    return frames


def settle(torch, pipe, frames, dev, window=10, tol=0.01, cap=100):
    """Untimed settle loop before the W warm-ups: after an idle period (pipeline construction, synthetic frames) the first
    launches of a ~1 ms pass run inside the chip's idle -> load power transient (profiles/r02/launch_drift.txt), and W = 5 steps
    are 5 ms.  Windows of ``window`` steps are repeated until two consecutive ones agree to ``tol`` (at most ``cap`` steps).
    Per rank, no collective.  Returns (steps run, ms per step of the last window)."""
    run, prev, cur = 0, None, None
    while run < cap:
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(window):
            pipe.step(frames)
        torch.cuda.synchronize(dev)
        cur = (time.perf_counter() - t0) / window * 1e3
        run += window
        if prev is not None and abs(cur - prev) <= tol * prev:
            break
        prev = cur
    return run, cur


def timed_steps(torch, D, pipe, frames, steps, warmup, dev):
    """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; MAX over ranks."""
    for _ in range(warmup):
        pipe.step(frames)
    torch.cuda.synchronize(dev)
    D.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.step(frames)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    D.barrier()
    timed_steps.own = elapsed                       # this rank's own time (the scaling record lists every rank's)
    return D.max_over_ranks(elapsed)


def dominant_kernel(torch, pipe, frames, wl, B, dev, launches=24):
    """Average launch duration of the dominant kernel from HIP events, in a loop of its own AFTER the timed region: the
    library brackets gray_stream_kernel (gray) / rgb_line_end2_kernel (rgb) with an event pair on the stream it launches on
    (silent_set_profiling / silent_profile_elapsed_ms) while the ordinary step runs -- i.e. the very instantiation the step uses."""
    # (the kernel is priced ALONE on the chip: with overlap=True the next batch's pyramid kernel runs beside it)
    overlapped = getattr(pipe, "overlap", False)
    if overlapped:
        torch.cuda.synchronize(dev)
        pipe.overlap = False
    pipe.set_profiling(1)
    ms, px = [], 0
    for _ in range(launches // 8):
        for _ in range(8):                       # the library keeps a ring of 8 event pairs
            pipe.step(frames)
        torch.cuda.synchronize(dev)
        t, px = pipe.profiled_kernel()
        ms.append(float(t))
        pipe.set_profiling(1)                    # restart the ring
    pipe.set_profiling(0)
    if overlapped:
        torch.cuda.synchronize(dev)
        pipe.overlap = True
    dom_ms = float(np.mean(ms))
    if wl["mode"] == "gray":
        other_px = pipe.frame_px * B - px
        # per level-0 pixel: frame read (4 B), pyramid + CS written (4 + 4), K end maps (4K); plus the pyramid of every
        # other level written once (4 B per pixel of those levels)
        nbytes = px * (4 + 4 + 4 + 4 * wl["n_orient"]) + other_px * 4
        return {"kernel": "gray_stream_kernel<%d>" % wl["n_orient"], "ms": dom_ms, "bytes": int(nbytes),
                "unit_level_pixels_per_launch": int(px), "pmc_name": pipe.dominant_kernel_name()}
    # rgb: pyramid read once + every returned map written once (12 B/px each)
    return {"kernel": "rgb_line_end2_kernel", "ms": dom_ms, "bytes": int(pipe.filter_bytes_per_frame() * B),
            "pmc_name": "rgb_line_end2_kernel"}


def ingest_record(torch, pipe, frames, wl, B, dev, steps=20):
    """The ingest leg of the reference's per-frame path (recognition_testing.py:141-143: host frame -> float32 -> pyramid -> feed),
    outside the timed region: LineEndPipeline.step_host on batches that sit in PINNED host memory (where a capture driver leaves
    them), uint8 as a camera delivers them and float32 as the reference feeds them.  Reported: the host-to-device rate of the
    copy alone, frames/s of the whole pass including the upload for both dtypes (two batches in flight: the copy of batch n + 1
    overlaps the compute of batch n), the same from pageable NumPy arrays (one more host memcpy into the staging ring), and
    overlap_frac = (t_copy + t_compute - t_pipelined) / min(t_copy, t_compute): 1 = the shorter leg is fully hidden."""
    h, w = wl["hw"]
    c = 1 if wl["mode"] == "gray" else 3

    def wall(fn, n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n

    out = {"frames_per_step": B, "source": "pinned host memory, two alternating batches", "steps": steps}
    t_comp = wall(lambda i: pipe.step(frames), steps)
    for name, dtype in (("u8", torch.uint8), ("f32", torch.float32)):
        host = [(torch.randint(0, 256, (B, h, w, c), dtype=torch.uint8) if dtype == torch.uint8
                 else torch.randint(0, 256, (B, h, w, c), dtype=torch.uint8).to(torch.float32)).pin_memory() for _ in range(2)]
        devbuf = torch.empty((B, h, w, c), dtype=dtype, device=dev)
        wall(lambda i: devbuf.copy_(host[i & 1], non_blocking=True), 3)
        t_copy = wall(lambda i: devbuf.copy_(host[i & 1], non_blocking=True), steps)
        wall(lambda i: pipe.step_host(host[i & 1]), 4)
        t_pipe = wall(lambda i: pipe.step_host(host[i & 1]), steps)
        nbytes = host[0].numel() * host[0].element_size()
        out["h2d_GBs" if name == "u8" else "h2d_GBs_f32"] = round(nbytes / t_copy / 1e9, 2)
        out["frames_per_s_with_upload_" + name] = round(B / t_pipe, 1)
        out["ms_per_step_with_upload_" + name] = round(t_pipe * 1e3, 4)
        out["overlap_frac" if name == "u8" else "overlap_frac_f32"] = round((t_copy + t_comp - t_pipe) / min(t_copy, t_comp), 3)
        if name == "u8":
            pageable = [x.numpy().copy() for x in host]
            wall(lambda i: pipe.step_host(pageable[i & 1]), 3)
            out["frames_per_s_with_upload_u8_from_numpy"] = round(B / wall(lambda i: pipe.step_host(pageable[i & 1]), steps), 1)
        del host, devbuf
    out["frames_per_s_resident"] = round(B / t_comp, 1)
    out["ms_per_step_resident"] = round(t_comp * 1e3, 4)
    return out


def latency_record():
    """The reference's per-frame path (LineEndDisplayer.callback, recognition_testing.py:136-144: ONE camera frame per call -- host
    uint8 frame in, the six fetched tensors in host memory out), outside every timed region: wall time per call, p50 / p99, for
    the native displayer (silent_displayer_step: one library call, a replayed HIP graph per frame), the per-op path eager and with
    torch's graph capture, and the device time of the native frame (events around its graph: upload to download)."""
    from pysilent_amd.recognition_testing import LineEndDisplayer
    out = {}
    for name, (h, w) in (("640x480", (480, 640)), ("1920x1080", (1080, 1920))):
        frames = [np.random.default_rng(s_).integers(0, 256, (h, w, 3)).astype(np.uint8) for s_ in range(4)]
        rec = {"frame": "%s x 3 uint8 -> six float32 maps on the host" % name}
        for label, kw, n in (("native", {}, 300), ("eager", {"native": False}, 60), ("graph", {"native": False, "use_graph": True}, 60)):
            disp = LineEndDisplayer(**kw)
            for i in range(10):
                res = disp.callback(frames[i & 3])
            ts, busy = [], []
            for i in range(n):
                t0 = time.perf_counter()
                res = disp.callback(frames[i & 3])
                ts.append((time.perf_counter() - t0) * 1e3)
            rec["%s_ms_p50" % label] = round(float(np.percentile(ts, 50)), 4)
            rec["%s_ms_p99" % label] = round(float(np.percentile(ts, 99)), 4)
            if label == "native":
                # callback(copy=False): the raw views of two alternating pinned slots (valid until the second next frame) instead of
                # callback's default, arrays the caller may keep (a pinned slot nobody references any more; no copy either)
                ts = []
                for i in range(n):
                    t0 = time.perf_counter()
                    res = disp.callback(frames[i & 3], copy=False)
                    ts.append((time.perf_counter() - t0) * 1e3)
                rec["native_views_ms_p50"] = round(float(np.percentile(ts, 50)), 4)
                rec["native_views_ms_p99"] = round(float(np.percentile(ts, 99)), 4)
                # the device time of a frame: a loop of its own with HIP events around the graph (step(timing=True) synchronises the
                # stream; the wall-time loops poll the frame's completion word instead)
                fd = disp._native[1]
                for i in range(100):
                    fd.step(frames[i & 3], timing=True)
                    busy.append(fd.gpu_ms)
                rec["gpu_busy_ms"] = round(float(np.median(busy)), 4)
                rec["levels"] = len(res[1])
                rec["result_bytes"] = int(sum(np.stack(r).nbytes for r in res[1:]))
                # the same with the frame captured INTO the displayer's pinned buffer (no staging copy on the host)
                fd = disp._native[1]
                np.copyto(fd.frame_buffer, frames[0])
                ts = []
                for i in range(n):
                    t0 = time.perf_counter()
                    fd.step(fd.frame_buffer)
                    ts.append((time.perf_counter() - t0) * 1e3)
                rec["native_in_place_ms_p50"] = round(float(np.percentile(ts, 50)), 4)
                rec["native_in_place_ms_p99"] = round(float(np.percentile(ts, 99)), 4)
            del disp, res
        out[name] = rec
    return out


def roofline_of(dom, B):
    gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic_per_launch(dom["pmc_name"], B),
            "traffic_source": pmc_traffic_per_launch.source,
            "algorithmic_bytes_per_launch": dom["bytes"], "avg_launch_ms": round(dom["ms"], 4),
            "timing": "HIP events around the kernel's launches in a separate loop after the timed region"}
    if "unit_level_pixels_per_launch" in dom:
        roof["unit_level_pixels_per_launch"] = dom["unit_level_pixels_per_launch"]
    return roof


def side_workload(torch, D, name, local, dev, rank, world, label=None, **over):
    """Short run of another BASELINE config on this rank (N = 1 only): ms/step, whole-pass and dominant-kernel
    fractions of the HBM peak, in the same JSON line as the headline."""
    wl = WORKLOADS[name]
    B = wl["frames"]
    consts = D.broadcast_constants(wl["mode"], wl["n_orient"], device=local)
    overlap = over.pop("overlap", overlap_policy(world))
    pipe = make_pipeline(wl, B, local, consts, **over)
    frames = make_frames(torch, D, wl, B, rank, world, dev)
    torch.cuda.synchronize(dev)
    tune_pipeline(pipe, frames, overlap, placement_policy())
    # settle first: building the pipeline and the synthetic frames leaves the GPU idle for a second or two, and the
    # first ~20 launches after an idle period run inside the power-management transient (profiles/r02/launch_drift.txt)
    steps = 30
    settled, _ = settle(torch, pipe, frames, dev)
    elapsed = timed_steps(torch, D, pipe, frames, steps, 5, dev)
    dom = dominant_kernel(torch, pipe, frames, wl, B, dev, launches=16)
    h, w = wl["hw"]
    whole = pipe.algorithmic_bytes_per_frame() * B * steps / elapsed / 1e9
    out = {"workload": label or wl["name"], "frames_per_step": B, "settle_steps_run": settled,
           "ms_per_step": round(elapsed / steps * 1e3, 4),
           "mpx_in_per_s": round(B * steps * h * w / elapsed / 1e6, 1),
           "algorithmic_bytes_per_frame": pipe.algorithmic_bytes_per_frame(),
           "whole_pass_frac_of_hbm_peak": round(whole / HBM_PEAK_GBS, 4),
           "dominant_kernel": dom["kernel"], "dominant_kernel_ms": round(dom["ms"], 4),
           "dominant_kernel_frac_of_hbm_peak": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    out["streams"] = ("two (overlap): first half of batch n + 1 beside the second half of batch n" if pipe.overlap
                      else "one: every launch of a step back to back")
    if pipe.overlap_tuning:
        out["overlap_tuning"] = pipe.overlap_tuning
    out["placement_tuning"] = pipe.placement_tuning
    if wl["mode"] == "rgb":
        pipe.step(frames)
        pipe.wait()
        out["sparse_keypoint_tail"] = pipe.sparse_tail_stats()
        counts = pipe.kp_counts.cpu().numpy()
        out["keypoints_per_frame"] = round(float(counts.mean()), 1)
        out["max_keypoints_in_a_frame"] = int(counts.max())
        out["keypoint_capacity_per_frame"] = int(pipe.kp_cap)
        out["keypoints_truncated_frames"] = int((counts > pipe.kp_cap).sum())
        if out["keypoints_truncated_frames"]:
            raise RuntimeError("bench.py %s: %d frame(s) produced more keypoint rows than the capacity %d (max %d): rows were "
                               "dropped inside the timed region" % (name, out["keypoints_truncated_frames"], pipe.kp_cap, counts.max()))
    del pipe, frames
    torch.cuda.empty_cache()
    return out


def config3_variants(torch, D, local, dev, rank, world, cpu=True):
    """Config 3 on both output sets of SURVEY.md section 8d (line_end + keypoints; the same + the optional orientation map),
    on the bench's noise frames and -- the worst case for the sparse keypoint tail, which then hands most levels to the dense
    kernels -- with every frame a line drawing under the default 'ieee' policy."""
    out = {"config3": side_workload(torch, D, "config3", local, dev, rank, world),
           "config3_one_stream": side_workload(torch, D, "config3", local, dev, rank, world, overlap=False,
                                               label="config 3 on one stream (no overlap between consecutive steps)"),
           "config3_line_end_only": side_workload(torch, D, "config3", local, dev, rank, world,
                                                  label="config 3, SURVEY 8d output set: line_end + keypoints (no orientation map)",
                                                  orient_map=False),
           "config3_peak_value_map": side_workload(torch, D, "config3", local, dev, rank, world,
                                                   label="config 3 with the selection's peak-value map returned too (sparse tail + zero fill)",
                                                   peak_value_map=True)}
    if cpu:   # the host number beside config 3's GPU number (C port on every core + NumPy / SciPy on 2 frames)
        out["config3"]["cpu_baseline"] = cpu_baseline(WORKLOADS["config3"], D.broadcast_constants("rgb", 3, device=local))
    # the round-2 tail for comparison: the dense selection / count kernels on every level (SILENT_TUNE_RGB bit 5)
    from pysilent_amd import _lib, _runtime
    ctx = _runtime.get_context(local)
    old = ctx.get_tuning(_lib.TUNE_RGB)
    ctx.set_tuning(_lib.TUNE_RGB, old | 32)
    try:
        out["config3_dense_tail"] = side_workload(torch, D, "config3", local, dev, rank, world,
                                                  label="config 3 with the dense keypoint tail of round 2 (peak-value map through memory)",
                                                  peak_value_map=True)
    finally:
        ctx.set_tuning(_lib.TUNE_RGB, old)
    return out


SIDE_KEYS = ("config3", "config3_one_stream", "config3_line_end_only", "config3_peak_value_map", "config5", "reference_layout",
             "reference_layout_one_stream", "reference_layout_gray")


def record_side(out):
    """Every BASELINE config of the run where a reader who keeps only the headline keys (or only the tail of the line) still finds
    it: flat scalars in ``config`` -- side_<workload>_ms / _frac (whole pass, fraction of the 8 TB/s peak) / _first_draw_ms (the
    step on the FIRST allocation of the maps, before LineEndPipeline.tune_placement drew again) -- the same as ``config.side``
    {name: [ms, frac, first_draw_ms]}, and one compact string as the LAST key of the line (``summary``)."""
    cfg = out["config"]
    pt = cfg.get("placement_tuning") or {}
    cfg["first_draw_ms"] = (pt.get("tries_ms") or [None])[0]
    cfg["chosen_ms"] = pt.get("chosen_ms")
    side = {}
    for k in SIDE_KEYS:
        w = (out.get("other_workloads") or {}).get(k)
        if not w:
            continue
        first = ((w.get("placement_tuning") or {}).get("tries_ms") or [None])[0]
        side[k] = [w["ms_per_step"], w["whole_pass_frac_of_hbm_peak"], first]
        cfg["side_%s_ms" % k] = w["ms_per_step"]
        cfg["side_%s_frac" % k] = w["whole_pass_frac_of_hbm_peak"]
        cfg["side_%s_first_draw_ms" % k] = first
        cfg["side_%s_kernel_frac" % k] = w["dominant_kernel_frac_of_hbm_peak"]
    lat = ((out.get("latency") or {}).get("640x480") or {})
    if lat:
        cfg["latency_480p_ms"] = lat.get("native_ms_p50")
        cfg["latency_480p_views_ms"] = lat.get("native_views_ms_p50")
        cfg["latency_480p_in_place_ms"] = lat.get("native_in_place_ms_p50")
        cfg["latency_480p_gpu_busy_ms"] = lat.get("gpu_busy_ms")
    cfg["side"] = side
    parts = ["%s %.4f ms %.1f%% (first draw %s)" % (out["config"]["workload"].split(",")[0], out["ms_per_step"],
                                                     100 * cfg["whole_pass_frac_of_hbm_peak"], cfg["first_draw_ms"])]
    parts += ["%s %.4f ms %.1f%% (first draw %s)" % (k, v[0], 100 * v[1], v[2]) for k, v in side.items()]
    if lat:
        parts.append("latency 640x480 p50: %.4f ms callback (arrays the caller may keep), %.4f raw views, %.4f in place, GPU busy %.4f" % (
            lat.get("native_ms_p50", 0), lat.get("native_views_ms_p50", 0), lat.get("native_in_place_ms_p50", 0), lat.get("gpu_busy_ms", 0)))
    r = out["roofline"]
    parts.append("roofline %s %.4f ms frac %.4f traffic %s" % (r["kernel"], r["avg_launch_ms"], r["frac"], r["traffic"]))
    text = "SUMMARY whole pass ms/step, % of 8 TB/s: " + "; ".join(parts)
    out["summary"] = text[:1500]


_REAL_STDOUT = None


def quiet_stdout():
    """Rank 0 prints ONE JSON line: everything else that lands on file descriptor 1 (RCCL's version banner, gloo's connection
    notes, library chatter) is sent to stderr; emit_line() writes to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(text):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(text)
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, (text + "\n").encode())


def whole_job_mpx(frames_per_rank_per_step, world, steps, h, w, slowest_rank_seconds):
    """The contract's ``value``: input megapixels ALL ranks processed in the timed region / the time of the SLOWEST rank."""
    return frames_per_rank_per_step * world * steps * h * w / slowest_rank_seconds / 1e6


def dist_record(D, rank, local, ident, own_ms, **extra):
    """all_gather of one record per rank -> the ``dist`` object of the JSON line; exits non-zero (every rank) when two
    ranks report the same GPU: N ranks must have seen N distinct devices.  ``extra``: what this rank's tuners decided (streams,
    placement) and how long it settled."""
    rec = dict(rank=rank, local_rank=local, host=socket.gethostname(), pid=os.getpid(), ms_per_step=round(own_ms, 4), **ident)
    rec.update(extra)
    recs = sorted(D.gather_records(rec), key=lambda r: r["rank"])
    dup = D.duplicate_devices(recs) if os.environ.get("SILENT_BENCH_SHARE_GPU") != "1" else []
    out = {"backend": D.backend_name(), "world_size": len(recs), "distinct_devices": len({r["pci_bus_id"] for r in recs}),
           "ranks": recs}
    if dup:
        # (every rank says so: the launcher shows the log of whichever rank exits first)
        print("bench.py rank %d: ranks share a GPU: %s" % (rank, ", ".join("ranks %d and %d on %s" % d for d in dup)), file=sys.stderr)
        D.finalize()
        sys.exit(3)
    return out


def dry_run(args):
    """SILENT_BENCH_DRY=1 (the CPU test of the launcher): everything a rank does except the GPU work."""
    from pysilent_amd import distributed as D
    wl = WORKLOADS[args.workload]
    rank, world, local = D.init(backend=os.environ.get("SILENT_DIST_BACKEND", "gloo"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if os.environ.get("SILENT_BENCH_DRY_FAIL_RANK") == str(rank):
        print("dry run: rank %d fails on purpose" % rank, file=sys.stderr)
        sys.exit(7)
    if os.environ.get("SILENT_BENCH_DRY_HANG_RANK") == str(rank):
        print("dry run: rank %d (pid %d) hangs on purpose" % (rank, os.getpid()), file=sys.stderr, flush=True)
        while True:
            time.sleep(1.0)
    consts = D.broadcast_constants(wl["mode"], wl["n_orient"])
    mine = D.shard_frame_indices(4 * world, rank, world)
    D.barrier()
    slow = D.max_over_ranks(1.0 + rank)
    same = os.environ.get("SILENT_BENCH_DRY_SAME_BUS") == "1"
    ident = {"device_name": "dry-run (no GPU)", "pci_bus_id": "dry:%02d" % (0 if same else rank), "uuid": "", "gcn_arch": ""}
    dist = dist_record(D, rank, local, ident, 1.0 + rank, settle_steps_run=0, streams="one", overlap_policy=str(overlap_policy(world)),
                       placement_chosen_ms=None, placement_tries_ms=None)
    if rank == 0:
        h, w = wl["hw"]
        # every rank "processed" 4 frames per step in (1 + rank) seconds: value = SUM of frames over ranks / MAX time
        emit_line(json.dumps({"metric": METRIC, "value": whole_job_mpx(4, world, args.steps, h, w, slow), "dry_run": True,
                              "steps": args.steps, "frames_per_rank_per_step": 4, "frame_hw": [h, w],
                              "n_gpus": world, "slowest_rank_time": slow,
                              "frames_of_rank0": mine, "constants": sorted(consts), "dist": dist}))
    D.finalize()


def run_rank(args):
    wl = WORKLOADS[args.workload]
    quiet_stdout()
    if os.environ.get("SILENT_BENCH_DRY") == "1":
        return dry_run(args)

    import torch
    from pysilent_amd import distributed as D

    # rehearsal knobs (one-GPU box): SILENT_BENCH_SHARE_GPU=1 puts every rank on GPU 0, SILENT_DIST_BACKEND=gloo
    # moves the (init-only) collectives to the CPU; the driver's real multi-GPU runs use neither
    share = os.environ.get("SILENT_BENCH_SHARE_GPU") == "1"
    if share:
        os.environ["SILENT_DEVICE"] = "0"
    # a short box must say so before any collective does (device_count does not initialise the GPU)
    want = 1 if share else max(args.gpus, int(os.environ.get("LOCAL_RANK", "0")) + 1)
    have = torch.cuda.device_count()
    # a launcher that hands every rank ONE device through a visibility mask: that device is index 0 for this rank (the dist
    # record's PCI bus ids still prove N distinct devices)
    masked = have == 1 and args.gpus > 1 and any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    if have < want and not masked:
        print("bench.py: --gpus %d needs %d visible GPUs, torch.cuda.device_count() is %d" % (args.gpus, want, have), file=sys.stderr)
        sys.exit(4)
    if masked:
        os.environ["SILENT_DEVICE"] = "0"
    rank, world, local = D.init(backend=os.environ.get("SILENT_DIST_BACKEND"), device=0 if (share or masked) else None)
    if share or masked:
        local = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # one RCCL broadcast of the constant kernels at init; nothing crosses GPUs per frame
    consts = D.broadcast_constants(wl["mode"], wl["n_orient"], device=local)

    h, w = wl["hw"]
    c = 1 if wl["mode"] == "gray" else 3
    # working set per rank well beyond the 256 MiB Infinity Cache (hard part 6)
    B = args.frames or wl["frames"]
    pipe = make_pipeline(wl, B, local, consts)
    frames = make_frames(torch, D, wl, B, rank, world, dev)
    torch.cuda.synchronize(dev)
    tune_pipeline(pipe, frames, False if args.one_stream else overlap_policy(world), placement_policy() and not args.no_placement)
    # a marker in the kernel trace: everything the tuners launched (losing placements, candidate stream pairs) lies BEFORE this
    # trace_marker_kernel -- scripts/summarize_profile.py computes the per-kernel statistics over what follows it
    pipe.ctx.check(pipe._lib.silent_trace_marker_dev(pipe.ctx.handle, pipe._stream()))
    torch.cuda.synchronize(dev)

    settle_run, settle_ms = settle(torch, pipe, frames, dev)
    elapsed = timed_steps(torch, D, pipe, frames, args.steps, args.warmup, dev)
    own_ms = timed_steps.own / args.steps * 1e3

    # ---- everything below is outside the timed region ----
    # Steady state, reported beside `value`, never instead of it: after an idle period the first ~20 back-to-back launches
    # of the pass run up to 20 % slower than the rate the chip then settles at (profiles/r02/launch_drift.txt: 864 -> 1040
    # -> 848 us for the dominant kernel; flat with 20 ms idle gaps; a plain device copy does the same) -- a power-management
    # transient, and with --warmup 5 the K timed steps sit inside it.  60 more steps, timed the same way, show the settled rate.
    settle_steps = 60
    steady = timed_steps(torch, D, pipe, frames, settle_steps, 0, dev)
    dom = dominant_kernel(torch, pipe, frames, wl, B, dev)
    ingest = ingest_record(torch, pipe, frames, wl, B, dev) if world == 1 and not args.no_ingest else None
    latency = latency_record() if world == 1 and not args.no_latency else None
    # the scaling record proves itself: backend, and per rank the device it ran on and its own step time
    dist = dist_record(D, rank, local, D.device_identity(local), own_ms,
                       settle_steps_run=settle_run, streams="two" if pipe.overlap else "one",
                       overlap_policy=str(False if args.one_stream else overlap_policy(world)),
                       placement_chosen_ms=(pipe.placement_tuning or {}).get("chosen_ms"),
                       placement_tries_ms=(pipe.placement_tuning or {}).get("tries_ms"))
    if rank != 0:
        D.finalize()
        return
    total_frames = B * world * args.steps
    mpx_in = whole_job_mpx(B, world, args.steps, h, w, elapsed)
    whole = pipe.algorithmic_bytes_per_frame() * B * args.steps / elapsed / 1e9
    out = {
        "metric": METRIC if h == 1080 else "Mpx/s full pyramid line-end pass @4K",
        "value": round(mpx_in, 2),
        "unit": "Mpx/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "settle_steps_run": settle_run,
        "settle_last_window_ms_per_step": round(settle_ms, 4),
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl["name"], "frames_per_gpu_per_step": B, "frame": "%dx%dx%d f32" % (w, h, c),
                   "pyramid_px_per_frame": pipe.frame_px, "level_extents": pipe.extents,
                   "mpx_pyramid_per_s": round(total_frames * pipe.frame_px / elapsed / 1e6, 2),
                   "frames_per_s": round(total_frames / elapsed, 1),
                   "algorithmic_bytes_per_frame": pipe.algorithmic_bytes_per_frame(),
                   "whole_pass_algorithmic_GBs_per_gpu": round(whole, 1),
                   "whole_pass_frac_of_hbm_peak": round(whole / HBM_PEAK_GBS, 4),
                   # SURVEY.md section 8d quotes the fraction against the measured float4-copy peak as well
                   "whole_pass_frac_of_measured_copy_peak": round(whole / COPY_PEAK_GBS, 4),
                   "launches_per_step": pipe.launch_summary(),
                   "streams": ("two (overlap): first half of batch n + 1 beside the second half of batch n" if pipe.overlap
                               else "one: every launch of a step back to back"),
                   "overlap_tuning": pipe.overlap_tuning,
                   "overlap_policy": "SILENT_OVERLAP=%s -> %s" % (os.environ.get("SILENT_OVERLAP", "(unset)"), False if args.one_stream else overlap_policy(world)),
                   "placement_tuning": pipe.placement_tuning,
                   "sharding": "frame i -> rank i mod N; one RCCL broadcast of constants at init",
                   "csrc_revision": csrc_revision()},
        "roofline": roofline_of(dom, B),
        "ingest": ingest,
        "latency": latency,
        "dist": dist,
        "steady_state": {"ms_per_step": round(steady / settle_steps * 1e3, 4), "steps": settle_steps,
                         "value": round(B * world * settle_steps * h * w / steady / 1e6, 2),
                         "whole_pass_frac_of_hbm_peak": round(pipe.algorithmic_bytes_per_frame() * B * settle_steps / steady / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "cross-check of the settle loop: the same steps timed again right after the K timed ones "
                                 "(without a settle loop the first ~20 launches after an idle period run inside a power-management "
                                 "transient, profiles/r02/launch_drift.txt)"},
    }
    del pipe, frames
    torch.cuda.empty_cache()
    if world == 1 and not args.no_side_workloads:
        out["other_workloads"] = {}
        if args.workload != "config3":
            out["other_workloads"].update(config3_variants(torch, D, local, dev, rank, world, cpu=not args.no_cpu_baseline))
        if args.workload != "config5":
            out["other_workloads"]["config5"] = side_workload(torch, D, "config5", local, dev, rank, world)
        if args.workload != "reference_layout":
            out["other_workloads"]["reference_layout"] = side_workload(torch, D, "reference_layout", local, dev, rank, world)
            out["other_workloads"]["reference_layout_gray"] = side_workload(torch, D, "reference_layout_gray", local, dev, rank, world)
            out["other_workloads"]["reference_layout_one_stream"] = side_workload(
                torch, D, "reference_layout", local, dev, rank, world, overlap=False,
                label="the reference's layout on one stream (no overlap between consecutive steps)")
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, consts)
    else:
        out["cpu_baseline"] = None
    record_side(out)
    emit_line(json.dumps(out))
    # the same summary once more as the LAST thing on stderr: a reader who keeps only the tail of the output still sees every config
    print(out["summary"], file=sys.stderr, flush=True)
    D.finalize()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=0, help="frames per rank per step (0 = workload default)")
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-workloads", action="store_true")
    ap.add_argument("--no-ingest", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-placement", action="store_true", help="keep the first allocation of the maps (no tune_placement)")
    ap.add_argument("--one-stream", action="store_true",
                    help="every launch of a step back to back on one stream (no overlap between consecutive steps); what the rocprofv3 "
                         "per-kernel traces are taken with -- overlapped kernels stretch each other's durations")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
