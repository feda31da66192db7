#!/usr/bin/env python3
"""Whole-step time of a workload (LineEndPipeline.step, HIP events around 10 steps, 6 rounds) for the library selected by
SILENT_LIB_PATH: python scripts/ab_steps.py config2|config3|config5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "config2"]
from pysilent_amd.pipeline import LineEndPipeline
B = wl["frames"]
kw = dict(mode=wl["mode"], n_levels=wl["n_levels"], batch=B, device=0)
if wl["mode"] == "gray":
    kw["n_orient"] = wl["n_orient"]
else:
    kw.update(max_keypoints_per_frame=1 << 16, selection=True, value_map=False)
pipe = LineEndPipeline(wl["hw"], **kw)
c = 1 if wl["mode"] == "gray" else 3
shape = (B,) + wl["hw"] + (c,)
frames = torch.randint(0, 256, shape, device="cuda").float()
for _ in range(30):
    pipe.step(frames)
torch.cuda.synchronize()
ts = []
for rnd in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        pipe.step(frames)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 10)
print("%s step: median %.4f ms  min %.4f" % (sys.argv[1] if len(sys.argv) > 1 else "config2", np.median(ts), np.min(ts)))
