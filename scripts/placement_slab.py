#!/usr/bin/env python3
"""Follow-up of placement_probe.py: every buffer of a gray workload carved out of ONE device allocation, at controlled relative
offsets.  Question 1: with one allocation, does the kernel time still move from allocation to allocation?  Question 2: does it move
with the relative offsets of the buffers (extra padding between them)?

    python scripts/placement_slab.py config5 [rounds]
"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from pysilent_amd.pipeline import LineEndPipeline

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = bench.WORKLOADS[name]
B = wl["frames"]
kw = dict(mode=wl["mode"], n_levels=wl["n_levels"], batch=B, device=0, n_orient=wl["n_orient"])


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def measure(pipe, frames):
    for _ in range(40):
        pipe.step(frames)
    torch.cuda.synchronize()
    step = float(np.median([timed(lambda: pipe.step(frames), 10) for _ in range(4)]))
    pipe.set_profiling(1)
    for _ in range(8):
        pipe.step(frames)
    torch.cuda.synchronize()
    kern = pipe.profiled_kernel()[0]
    pipe.set_profiling(0)
    return step, kern


pipe = LineEndPipeline(wl["hw"], **kw)
n = pipe.batch * pipe.frame_px
sizes = {"frames": B * wl["hw"][0] * wl["hw"][1], "pyr": n, "cs": n, "end": n * pipe.n_orient}      # floats
del pipe.cs, pipe.end
pipe._pyrs = []
pipe.pyr = None
gc.collect()
torch.cuda.empty_cache()
MB = 1 << 20
pad_sets = [(0, 0, 0), (0, 0, 0), (4096, 8192, 12288), (65536, 131072, 196608), (256, 512, 768), (1 * MB, 2 * MB + 4096, 3 * MB + 8192),
            (0, 0, 0), (2 * MB, 2 * MB, 2 * MB)]
for rnd in range(rounds):
    for pads in pad_sets:
        gc.collect()
        torch.cuda.empty_cache()
        total = sum(sizes.values()) * 4 + sum(pads) + 16 * MB
        slab = torch.empty(total, dtype=torch.uint8, device="cuda")
        off = (-slab.data_ptr()) % (2 * MB)              # 2 MiB aligned start
        views = {}
        for (nm, cnt), pad in zip(sizes.items(), (0,) + tuple(pads)):
            off += pad
            views[nm] = slab[off:off + cnt * 4].view(torch.float32)
            off += cnt * 4
            off += (-off) % 256
        frames = views["frames"].view((B,) + wl["hw"] + (1,))
        frames.copy_(torch.randint(0, 256, frames.shape, device="cuda").float())
        pipe.pyr, pipe.cs, pipe.end = views["pyr"], views["cs"], views["end"]
        pipe._pyrs = [pipe.pyr]
        step, kern = measure(pipe, frames)
        print("round %d pads %-28s step %.4f ms  kernel %.4f ms   slab@%x" % (rnd, pads, step, kern, slab.data_ptr()), flush=True)
        del slab, views, frames
        pipe.pyr = pipe.cs = pipe.end = None
        pipe._pyrs = []
