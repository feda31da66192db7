#!/usr/bin/env python3
"""Fifth placement experiment: does the SIZE of the request steer the driver's choice of pages?  Only the big K-orientation map
matters for config 5 (placement_which.py); it is re-allocated as (a) its exact size, (b) the next power of two, (c) a multiple of
1 GiB, (d) carved from one 32 GiB arena at varying offsets -- six times each, the others held.
    python scripts/placement_pow2.py config5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
frames = torch.randint(0, 256, (B,) + wl["hw"] + (1,), device="cuda").float()
n = pipe.end.numel()


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
print("base: kernel %.4f ms, end = %.2f GiB" % (kernel_ms(), n * 4 / 2 ** 30), flush=True)
held = []
GiB = 1 << 30
sizes = {"exact": n * 4, "next power of two": 1 << (n * 4 - 1).bit_length(), "multiple of 1 GiB": -(-n * 4 // GiB) * GiB, "exact + 2 MiB": n * 4 + (2 << 20)}
for label, nbytes in sizes.items():
    line = []
    for t in range(6):
        raw = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        held.append(raw)
        pipe.end = raw[:n * 4].view(torch.float32)
        line.append("%.4f" % kernel_ms())
    print("%-20s (%6.2f GiB) -> kernel ms  %s" % (label, nbytes / GiB, "  ".join(line)), flush=True)
    held.clear()
    torch.cuda.empty_cache()
arena = torch.empty(32 * GiB, dtype=torch.uint8, device="cuda")
line = []
for k in range(6):
    off = k * 5 * GiB
    pipe.end = arena[off:off + n * 4].view(torch.float32)
    line.append("%.4f" % kernel_ms())
print("one 32 GiB arena, offsets 0, 5, 10 .. GiB -> kernel ms  %s" % "  ".join(line), flush=True)
del arena
pipe.end = None
torch.cuda.empty_cache()
for gib in (6, 8, 12, 16, 24, 32, 64):
    line = []
    held = []
    for t in range(4):
        a = torch.empty(gib * GiB, dtype=torch.uint8, device="cuda")
        held.append(a)
        pipe.end = a[:n * 4].view(torch.float32)
        line.append("%.4f" % kernel_ms())
        pipe.end = a[gib * GiB - n * 4:].view(torch.float32)
        line.append("(tail %.4f)" % kernel_ms())
    print("arena of %2d GiB, allocated 4 times (all held): kernel ms  %s" % (gib, "  ".join(line)), flush=True)
    pipe.end = None
    held.clear()
    torch.cuda.empty_cache()
