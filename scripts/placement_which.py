#!/usr/bin/env python3
"""Fourth placement experiment: which buffer's pages matter?  One base allocation of every buffer; then ONE buffer at a time is
re-allocated several times (the others stay) and the dominant kernel is timed.
    python scripts/placement_which.py config5 [tries]"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pysilent_amd.pipeline import LineEndPipeline

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
c = 1 if wl["mode"] == "gray" else 3
frames = torch.randint(0, 256, (B,) + wl["hw"] + (c,), device="cuda").float()
names = ("pyr", "cs", "end") if wl["mode"] == "gray" else ("pyr", "orient", "line_end")


def kernel_ms():
    for _ in range(12):
        pipe.step(frames)
    pipe.set_profiling(1)
    for _ in range(8):
        pipe.step(frames)
    torch.cuda.synchronize()
    t = pipe.profiled_kernel()[0]
    pipe.set_profiling(0)
    return t


for _ in range(40):
    pipe.step(frames)
print("base: kernel %.4f ms" % kernel_ms(), flush=True)
held = []
for nm in names + ("frames",):
    line = []
    for t in range(tries):
        if nm == "frames":
            new = torch.empty_like(frames)
            new.copy_(frames)
            held.append(frames)
            frames = new
        else:
            old = pipe._pyrs[0] if nm == "pyr" else getattr(pipe, nm)
            new = torch.empty_like(old)
            held.append(old)
            if nm == "pyr":
                pipe._pyrs[0] = pipe.pyr = new
            else:
                setattr(pipe, nm, new)
        line.append("%.4f" % kernel_ms())
    print("re-allocating only %-8s -> kernel ms  %s" % (nm, "  ".join(line)), flush=True)
