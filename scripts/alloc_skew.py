#!/usr/bin/env python3
"""Does the step time depend on where the pipeline's buffers land?  Re-create the pipeline behind a dummy allocation of
varying size (the caching allocator is emptied in between) and time the step: scripts/alloc_skew.py config5 [pads in KiB ...]"""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pysilent_amd.pipeline import LineEndPipeline

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
pads = [int(a) for a in sys.argv[2:]] or [0, 4, 64, 256, 1024, 2048, 3072, 8192, 0, 0]
wl = bench.WORKLOADS[name]
B = wl["frames"]
kw = dict(mode=wl["mode"], n_levels=wl["n_levels"], batch=B, device=0)
if wl["mode"] == "gray":
    kw["n_orient"] = wl["n_orient"]
else:
    kw.update(max_keypoints_per_frame=1 << 16, selection=True, value_map=False)
c = 1 if wl["mode"] == "gray" else 3
for pad in pads:
    gc.collect(); torch.cuda.empty_cache()
    dummy = torch.empty(max(pad, 0) * 1024 + 16, dtype=torch.uint8, device="cuda")
    pipe = LineEndPipeline(wl["hw"], **kw)
    frames = torch.randint(0, 256, (B,) + wl["hw"] + (c,), device="cuda").float()
    for _ in range(30):
        pipe.step(frames)
    torch.cuda.synchronize()
    ts = []
    for rnd in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            pipe.step(frames)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10)
    ptrs = {n: getattr(pipe, n).data_ptr() for n in ("pyr", "cs", "end") if hasattr(pipe, n) and getattr(pipe, n) is not None}
    print("pad %6d KiB  step median %.4f ms  min %.4f   frames %x  %s" % (pad, np.median(ts), np.min(ts), frames.data_ptr(),
          " ".join("%s %x" % kv for kv in ptrs.items())))
    del pipe, frames, dummy
