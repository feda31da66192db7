#!/usr/bin/env python3
"""Why does the same kernel on the same box take 1.21 ... 1.56 ms from one process to the next ("measurement hygiene")?
Re-create a workload's buffers several times IN ONE PROCESS (same virtual addresses after empty_cache, fresh physical pages) and
print, per allocation, the settled step, the dominant kernel's time and the plain copy rate of every buffer on its own -- if the
copy rates move with the kernel time, the spread is a property of where the pages landed, not of the kernel.

    python scripts/placement_probe.py config5 [rounds] [--hold]      # --hold: keep every earlier allocation alive (new VA, new pages)
"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from pysilent_amd.pipeline import LineEndPipeline

args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "config5"
rounds = int(args[1]) if len(args) > 1 else 8
hold = "--hold" in sys.argv
wl = bench.WORKLOADS[name]
B = wl["frames"]
kw = dict(mode=wl["mode"], n_levels=wl["n_levels"], batch=B, device=0)
if wl["mode"] == "gray":
    kw["n_orient"] = wl["n_orient"]
else:
    kw.update(selection=True, value_map=False, peak_value_map=False)
c = 1 if wl["mode"] == "gray" else 3


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def copy_rate(t):
    """GB/s of copying the first half of a buffer onto its second half (reads + writes)."""
    flat = t.view(-1)
    h = flat.numel() // 2
    src, dst = flat[:h], flat[h:2 * h]
    for _ in range(3):
        dst.copy_(src)
    ms = min(timed(lambda: dst.copy_(src), 5) for _ in range(3))
    return 2 * h * flat.element_size() / ms / 1e6


kept = []
for rnd in range(rounds):
    gc.collect()
    torch.cuda.empty_cache()
    pipe = LineEndPipeline(wl["hw"], **kw)
    frames = torch.randint(0, 256, (B,) + wl["hw"] + (c,), device="cuda").float()
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
    bufs = {"frames": frames, "pyr": pipe.pyr}
    for n in ("cs", "end", "orient", "line_end"):
        if getattr(pipe, n, None) is not None:
            bufs[n] = getattr(pipe, n)
    rates = {n: copy_rate(t) for n, t in bufs.items()}
    # (the copies overwrote the maps; frames too: refill them so that the next step sees noise again -- the timing does not care)
    print("alloc %2d  step %.4f ms  kernel %.4f ms   copy GB/s: %s   %s" % (
        rnd, step, kern, "  ".join("%s %4.0f" % (n, r) for n, r in rates.items()),
        " ".join("%s@%x" % (n, t.data_ptr()) for n, t in bufs.items())), flush=True)
    if hold:
        kept.append((pipe, frames))
    del pipe, frames, bufs
