#!/usr/bin/env python3
"""Chain kernel + keypoint tail of a bench workload against the chain's tile height (SILENT_TUNE_RGB bits 8-15 = th / 2), one
process, same buffers, alternating rounds; "auto" = the library's own cost model.
    python3 scripts/sweep_ref_th.py reference_layout [th,th,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "reference_layout"
ths = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "0,12,16,24,32,48,64,96,192").split(",")]
wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
frames = torch.randint(0, 256, (B,) + wl["hw"] + (3,), device="cuda").float()
pipe.tune_placement(frames)
pipe.run_pyramid(frames)
torch.cuda.synchronize()
times = {t: [] for t in ths}
kern = {t: [] for t in ths}
for rnd in range(7):
    for t in ths:
        pipe.ctx.set_tuning(1, (t // 2) << 8)
        for _ in range(3):
            pipe.run_filters_keypoints()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            pipe.run_filters_keypoints()
        b.record()
        torch.cuda.synchronize()
        pipe.set_profiling(1)
        for _ in range(8):
            pipe.run_filters_keypoints()
        torch.cuda.synchronize()
        k = pipe.profiled_kernel()[0]
        pipe.set_profiling(0)
        if rnd >= 1:
            times[t].append(a.elapsed_time(b) / 10)
            kern[t].append(k)
pipe.ctx.set_tuning(1, 0)
for t in ths:
    print("%s th %-5s chain + tail median %.4f ms (min %.4f)   chain kernel %.4f" % (name, "auto" if t == 0 else t, np.median(times[t]), np.min(times[t]), np.median(kern[t])))
