#!/usr/bin/env python3
"""Third placement experiment: ONE allocation of every buffer (with slack), then one buffer at a time is shifted inside its own
allocation by small byte offsets while everything else stays put.  If the kernel time moves with the shift, the spread between
allocations is the relative PHASE of the streams on the memory channels (tunable per allocation); if it does not, it is something
coarser than an offset can reach.

    python scripts/placement_phase.py config5
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from pysilent_amd.pipeline import LineEndPipeline

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = LineEndPipeline(wl["hw"], mode="gray", n_levels=wl["n_levels"], batch=B, device=0, n_orient=wl["n_orient"])
n = pipe.batch * pipe.frame_px
SLACK = 1 << 22          # bytes
frames = torch.randint(0, 256, (B,) + wl["hw"] + (1,), device="cuda").float()
raw = {"pyr": torch.empty(n * 4 + SLACK, dtype=torch.uint8, device="cuda"), "cs": torch.empty(n * 4 + SLACK, dtype=torch.uint8, device="cuda"),
       "end": torch.empty(n * 4 * pipe.n_orient + SLACK, dtype=torch.uint8, device="cuda")}
counts = {"pyr": n, "cs": n, "end": n * pipe.n_orient}


def view(nm, off):
    return raw[nm][off:off + counts[nm] * 4].view(torch.float32)


def kernel_ms():
    for _ in range(6):
        pipe.step(frames)
    pipe.set_profiling(1)
    for _ in range(8):
        pipe.step(frames)
    torch.cuda.synchronize()
    t = pipe.profiled_kernel()[0]
    pipe.set_profiling(0)
    return t


offs = {"pyr": 0, "cs": 0, "end": 0}
pipe._adopt_maps({k: view(k, 0) for k in raw})
for _ in range(40):
    pipe.step(frames)
print("base kernel %.4f ms" % kernel_ms(), flush=True)
for nm in ("cs", "pyr", "end"):
    line = []
    for off in (0, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 18, 1 << 20, 1 << 21, 0):
        m = {k: view(k, offs[k]) for k in raw}
        m[nm] = view(nm, off)
        pipe._adopt_maps(m)
        line.append("%d: %.4f" % (off, kernel_ms()))
    print("%-4s shifted by bytes -> kernel ms   %s" % (nm, "   ".join(line)), flush=True)
