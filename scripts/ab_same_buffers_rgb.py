#!/usr/bin/env python3
"""scripts/ab_same_buffers.py for the RGB workloads, by parts: pyramid alone, chain + keypoint tail alone, whole step -- every
build on the SAME frames, pyramid and maps, alternating in one process.
    python scripts/ab_same_buffers_rgb.py <config3|reference_layout> libA.so libB.so [...]   [ROUNDS=6] [STEPS=20] [ALLOCS=2]"""
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
pipes = []
for path in libs:
    _lib._lib = None
    _lib.LIB_PATH = path
    _runtime._contexts.clear()
    pipes.append(bench.make_pipeline(wl, B, 0, None))
frames = torch.randint(0, 256, (B,) + wl["hw"] + (3,), device="cuda").float()


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
    keep.append(maps)
    for p in pipes:
        p._adopt_maps(maps)
        p.kp_idx, p.kp_counts = pipes[0].kp_idx, pipes[0].kp_counts
        for _ in range(10):
            p.step(frames)
    torch.cuda.synchronize()
    res = {i: {"pyramid": [], "chain+tail": [], "step": []} for i in range(len(pipes))}
    for r in range(rounds):
        for i, p in enumerate(pipes):
            for _ in range(3):
                p.step(frames)
            res[i]["pyramid"].append(timed(lambda: p.run_pyramid(frames), steps))
            res[i]["chain+tail"].append(timed(p.run_filters_keypoints, steps))
            res[i]["step"].append(timed(lambda: p.step(frames), steps))
    for i, path in enumerate(libs):
        print("alloc %d  %-24s %s" % (al, os.path.basename(path), "   ".join("%s %.4f (min %.4f)" % (k, np.median(v), np.min(v)) for k, v in res[i].items())), flush=True)
sys.stdout.flush()
os._exit(0)
