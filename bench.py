#!/usr/bin/env python3
"""bench.py -- Mpx/s of the full pyramid line-end pass @1080p (BASELINE.json metric), one rank per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B] [--workload config2|config3|config5]

A "step" is one pass of the hot path over one batch of B device-resident synthetic frames per rank:
zoom pyramid -> center-surround -> ReLU -> oriented line-end bank -> ReLU -> clip (config 2/5) or the
reference's RGB chain + keypoints (config 3).  Frames shard over ranks (frame i -> rank i mod N, no
per-frame collective); the constant kernels are generated on rank 0 and broadcast once over RCCL.

Rank 0 prints ONE JSON line.  ``value`` = input-frame megapixels per second over all ranks, inputs
resident in HBM when the timed region starts.  ``roofline`` prices the dominant kernel (the fused filter
pass) against HBM; ``cpu_baseline`` is the C port of the oracle timed on the host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: 1080p grayscale, 5-level pyramid, center-surround + 4-orientation line-end
    "config2": dict(hw=(1080, 1920), mode="gray", n_levels=5, n_orient=4,
                    name="1080p gray, 5-level pyramid (scale 2), CS + 4-orientation line-end"),
    # configs[2]: 1080p RGB, 6-level pyramid, normalize + peak extraction
    "config3": dict(hw=(1080, 1920), mode="rgb", n_levels=6, n_orient=3,
                    name="1080p RGB, 6-level pyramid (scale 2), rgc>rgby>stripe>regulate>end>pad>value, top 10 %, NMS, keypoints"),
    # configs[4]: 4K, 8-level pyramid, 8-orientation bank
    "config5": dict(hw=(2160, 3840), mode="gray", n_levels=8, n_orient=8,
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


def pmc_traffic_per_launch(kernel_substr, frames_in_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/<latest>/pmc_hbm_bytes.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench at 64
    frames per launch), corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts 64 B per
    128-B request -> x2; WRITE_SIZE exact; both in KiB.  Scaled linearly to this run's frames per launch."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_hbm_bytes.json")))
    if not cands:
        return None
    for path in reversed(cands):  # latest profile that holds this kernel
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        for name, c in prof.items():
            if kernel_substr in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                per64 = (2.0 * c["FETCH_SIZE"]["mean_KiB_per_dispatch"] + c["WRITE_SIZE"]["mean_KiB_per_dispatch"]) * 1024.0
                per64 *= 64.0 / c.get("frames_per_dispatch", 64)
                return int(per64 * frames_in_launch / 64.0)
    return None


def cpu_baseline(wl, consts, budget_s=12.0):
    """Time the C port of the oracle (oracle/silent_oracle.c, OpenMP over the host cores) on a bounded sample
    of the same workload.  The oracle is the thing timed here, never part of the GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle as co
    import silent_oracle as so
    from pysilent_amd.distributed import synthetic_frame
    h, w = wl["hw"]
    if wl["mode"] != "gray":
        return None
    extents = so.classic_extents(h, w, 2.0, wl["n_levels"])

    def one(i):
        frame = synthetic_frame(i, h, w, 1)
        t = time.perf_counter()
        pyr = co.classic_pyramid(frame, extents)
        for lev in pyr:
            co.gray_line_end_level(lev, consts["cs"], consts["end"])
        return time.perf_counter() - t

    one(0)                       # warm (page faults, OpenMP team start)
    n, total = 0, 0.0
    while (total < budget_s and n < 4096) or n < 4:     # bounded sample: ~12 s of CPU work
        total += one(1 + n)
        n += 1
    return {"value": round(n * h * w / total / 1e6, 3), "unit": "Mpx/s", "cores": co.num_threads(), "kind": "port",
            "sample": "%d synthetic %dx%d frames, whole pass (pyramid + CS + %d-orientation line-end), "
                      "oracle/silent_oracle.c -O3 -fopenmp float64 accumulation, %.1f s of CPU work"
                      % (n, w, h, wl["n_orient"], total),
            "host_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=0, help="frames per rank per step (0 = workload default)")
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]

    import torch
    from pysilent_amd import distributed as D
    from pysilent_amd.pipeline import LineEndPipeline

    # rehearsal knobs (one-GPU box): SILENT_BENCH_SHARE_GPU=1 puts every rank on GPU 0, SILENT_DIST_BACKEND=gloo
    # moves the (init-only) collectives to the CPU; the driver's real multi-GPU runs use neither
    share = os.environ.get("SILENT_BENCH_SHARE_GPU") == "1"
    if share:
        os.environ["SILENT_DEVICE"] = "0"
    rank, world, local = D.init(backend=os.environ.get("SILENT_DIST_BACKEND"), device=0 if share else None)
    if share:
        local = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # one RCCL broadcast of the constant kernels at init; nothing crosses GPUs per frame
    consts = D.broadcast_constants(wl["mode"], wl["n_orient"], device=local)

    h, w = wl["hw"]
    c = 1 if wl["mode"] == "gray" else 3
    # working set per rank well beyond the 256 MiB Infinity Cache (hard part 6)
    B = args.frames or {"config2": 64, "config3": 32, "config5": 16}[args.workload]
    pipe = LineEndPipeline((h, w), mode=wl["mode"], n_levels=wl["n_levels"], n_orient=wl["n_orient"], batch=B,
                           device=local, constants=consts, max_keypoints_per_frame=1 << 16,
                           **({"selection": True} if wl["mode"] == "rgb" else {}))
    frames = torch.empty((B, h, w, c), dtype=torch.float32, device=dev)
    for j, gi in enumerate(D.shard_frame_indices(B * world, rank, world)):
        frames[j] = torch.from_numpy(D.synthetic_frame(gi, h, w, c)).to(dev)
    torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        pipe.step(frames)
    torch.cuda.synchronize(dev)
    D.barrier()

    gray = wl["mode"] == "gray"
    if gray:
        # the dominant kernel (fused pyramid + CS + line-end for the unit-zoom level) is bracketed by HIP events
        # INSIDE the library, on the stream it is launched on (silent_set_profiling / silent_profile_elapsed_ms)
        # ... on a few steps of the timed region only (an event pair per step costs ~5 % of this 1 ms pass)
        pipe.set_profiling(max(1, args.steps // 6))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
           torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    dom_ms = []
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        a, b, e = ev[k]
        if gray:
            pipe.step(frames)                 # silent_gray_pass_dev: 3 launches, events around the dominant one
        else:
            a.record()
            pipe.run_pyramid(frames)
            b.record()
            pipe.run_filters()
            e.record()
            pipe.run_keypoints()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    D.barrier()
    elapsed = D.max_over_ranks(elapsed)

    if gray:
        # per-launch duration of the dominant kernel: mean over the event pairs recorded inside the timed region
        dom_ms, dom_px = pipe.profiled_kernel()
        dom_ms = float(dom_ms)
        pipe.set_profiling(0)
        pyr_ms = filt_ms = None
    else:
        pyr_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in ev]))
        filt_ms = float(np.mean([b.elapsed_time(e) for _, b, e in ev]))

    if rank != 0:
        D.finalize()
        return
    total_frames = B * world * args.steps
    mpx_in = total_frames * h * w / elapsed / 1e6
    mpx_pyr = total_frames * pipe.frame_px / elapsed / 1e6
    if gray:
        # dominant kernel = gray_stream_kernel: per level-0 pixel it reads the frame (4 B) and writes pyramid (4), CS (4)
        # and K end maps (4K); it also writes the pyramid of every other level (4 B per pixel of those levels)
        other_px = (pipe.frame_px * B) - dom_px
        dom_bytes = dom_px * (4 + 4 + 4 + 4 * wl["n_orient"]) + other_px * 4
        kname = "gray_stream_kernel<%d," % wl["n_orient"]
        roof = {"bound": "hbm", "kernel": "gray_stream_kernel<%d>" % wl["n_orient"],
                "achieved": round(dom_bytes / (dom_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic_per_launch(kname, B),
                "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4),
                "unit_level_pixels_per_launch": int(dom_px)}
    else:
        filt_bytes = pipe.filter_bytes_per_frame() * B
        roof = {"bound": "hbm", "kernel": "rgb_line_end_kernel", "achieved": round(filt_bytes / (filt_ms * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(filt_bytes / (filt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic_per_launch("rgb_line_end_kernel", B),
                "algorithmic_bytes_per_launch": filt_bytes, "avg_launch_ms": round(filt_ms, 4)}
    out = {
        "metric": METRIC if h == 1080 else "Mpx/s full pyramid line-end pass @4K",
        "value": round(mpx_in, 2),
        "unit": "Mpx/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl["name"], "frames_per_gpu_per_step": B, "frame": "%dx%dx%d f32" % (w, h, c),
                   "pyramid_px_per_frame": pipe.frame_px, "level_extents": pipe.extents,
                   "mpx_pyramid_per_s": round(mpx_pyr, 2), "frames_per_s": round(total_frames / elapsed, 1),
                   "algorithmic_bytes_per_frame": pipe.algorithmic_bytes_per_frame(),
                   "whole_pass_algorithmic_GBs_per_gpu": round(pipe.algorithmic_bytes_per_frame() * B * args.steps
                                                               / elapsed / 1e9, 1),
                   "whole_pass_frac_of_hbm_peak": round(pipe.algorithmic_bytes_per_frame() * B * args.steps
                                                        / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                   # SURVEY.md section 8d quotes the fraction against the measured float4-copy peak as well (6.29 TB/s,
                   # /opt/skills/guides/MI355X_MICROARCH.md)
                   "whole_pass_frac_of_measured_copy_peak": round(pipe.algorithmic_bytes_per_frame() * B * args.steps
                                                                  / elapsed / 1e9 / 6290.0, 4),
                   "launches_per_step": "gray_stream_kernel (whole pyramid + level-0 CS/line-end, frame read once) + "
                                        "gray_line_end_kernel (levels >= 1)"
                   if gray else "unit + region pyramid, fused RGB chain, max/min + fused selection (top 10 % > NMS > value), "
                                   "cell-max / count / scan / write keypoint kernels",
                   "pyramid_kernels_ms": None if pyr_ms is None else round(pyr_ms, 4),
                   "filter_kernel_ms": None if filt_ms is None else round(filt_ms, 4),
                   "sharding": "frame i -> rank i mod N; one RCCL broadcast of constants at init"},
        "roofline": roof,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, consts)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    D.finalize()


if __name__ == "__main__":
    main()
