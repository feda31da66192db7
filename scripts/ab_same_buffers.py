#!/usr/bin/env python3
"""A/B of library builds IN ONE PROCESS ON THE SAME BUFFERS.  The physical pages an allocation lands on move the same kernel by up
to 20 % (profiles/r05/placement.md), so two processes -- two allocations -- cannot tell a 3 % kernel change from luck.  Here every
build is loaded side by side (ctypes handles of their own, a context and a plan each), all pipelines adopt ONE set of maps and read
ONE batch of frames, and the timing windows alternate between the builds.

    python scripts/ab_same_buffers.py <workload> libA.so libB.so [...]   [ROUNDS=6] [STEPS=20] [ALLOCS=2]

ALLOCS: the whole comparison is repeated on that many fresh allocations (the ranking must hold on each).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from pysilent_amd import _lib, _runtime

name = sys.argv[1]
libs = [os.path.abspath(p) for p in sys.argv[2:]]
rounds, steps, allocs = int(os.environ.get("ROUNDS", "6")), int(os.environ.get("STEPS", "20")), int(os.environ.get("ALLOCS", "2"))
wl = bench.WORKLOADS[name]
B = wl["frames"]
c = 1 if wl["mode"] == "gray" else 3

pipes = []
for path in libs:
    _lib._lib = None                     # the binding caches ONE handle: load the next build as the current one ...
    _lib.LIB_PATH = path
    _runtime._contexts.clear()           # ... with a context of its own
    pipes.append(bench.make_pipeline(wl, B, 0, None))
frames = torch.randint(0, 256, (B,) + wl["hw"] + (c,), device="cuda").float()


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


keep = []
for al in range(allocs):
    maps = pipes[0]._alloc_maps()
    keep.append(maps)                    # (held: the next round gets other pages)
    for p in pipes:
        p._adopt_maps(maps)
        for _ in range(10):
            p.step(frames)
    torch.cuda.synchronize()
    res = {i: {"step": [], "kernel": []} for i in range(len(pipes))}
    for r in range(rounds):
        for i, p in enumerate(pipes):
            for _ in range(3):
                p.step(frames)
            res[i]["step"].append(timed(lambda: p.step(frames), steps))
            p.set_profiling(1)
            for _ in range(8):
                p.step(frames)
            torch.cuda.synchronize()
            res[i]["kernel"].append(p.profiled_kernel()[0])
            p.set_profiling(0)
    if os.environ.get("CHECK") == "1":     # the builds must write the same bits into the same maps
        names = [k for k in maps if maps[k] is not None]
        ref = None
        for i, p in enumerate(pipes):
            for k in names:
                maps[k].fill_(-7.0)
            p.step(frames)
            torch.cuda.synchronize()
            got = [maps[k].clone() for k in names]
            if ref is None:
                ref = got
            else:
                same = [bool(torch.equal(a.view(torch.int32), b.view(torch.int32))) for a, b in zip(ref, got)]
                print("alloc %d  %s writes the same bits as %s: %s" % (al, os.path.basename(libs[i]), os.path.basename(libs[0]),
                                                                         dict(zip(names, same))), flush=True)
            del got
        del ref
    for i, path in enumerate(libs):
        print("alloc %d  %-28s step %.4f (min %.4f)   dominant kernel %.4f (min %.4f) ms" % (
            al, os.path.basename(path), np.median(res[i]["step"]), np.min(res[i]["step"]), np.median(res[i]["kernel"]), np.min(res[i]["kernel"])), flush=True)
sys.stdout.flush()
os._exit(0)                              # (no destructors: plans and contexts belong to different builds of the library)
