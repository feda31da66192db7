#!/usr/bin/env python3
"""Config 3: one pipeline over the whole batch on one stream against the batch split over S pipelines (a private context
and a HIP stream each): do pyramid / chain / tail of different frame groups overlap usefully?
    python scripts/ab_two_streams.py [frames=32] [splits=1,2,4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pysilent_amd import _runtime
from pysilent_amd.pipeline import LineEndPipeline

wl = bench.WORKLOADS["config3"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else wl["frames"]
splits = [int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]
frames = torch.randint(0, 256, (B,) + wl["hw"] + (3,), device="cuda").float()


def build(S):
    parts = []
    for i in range(S):
        ctx = _runtime.Context(0)
        with _runtime.use_context(ctx):
            pipe = LineEndPipeline(wl["hw"], mode="rgb", n_levels=wl["n_levels"], batch=B // S, device=0, selection=True,
                                   value_map=False, peak_value_map=False, max_keypoints_per_frame=1 << 16)
        parts.append((ctx, pipe, torch.cuda.Stream(), frames[i * (B // S):(i + 1) * (B // S)]))
    return parts


def step(parts):
    cur = torch.cuda.current_stream()
    if len(parts) == 1:
        parts[0][1].step(parts[0][3])
        return
    ev = torch.cuda.Event()
    ev.record(cur)
    for ctx, pipe, st, fr in parts:
        st.wait_event(ev)
        with torch.cuda.stream(st):
            pipe.step(fr)
    for ctx, pipe, st, fr in parts:
        cur.wait_stream(st)


res = {}
built = {S: build(S) for S in splits}
for rnd in range(4):
    for S in splits:
        parts = built[S]
        for _ in range(10):
            step(parts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step(parts)
        torch.cuda.synchronize()
        res.setdefault(S, []).append((time.perf_counter() - t0) / 20 * 1e3)
for S in splits:
    print("splits %d: ms per %d frames: %s  median %.4f" % (S, B, " ".join("%.4f" % t for t in res[S]), np.median(res[S])))
