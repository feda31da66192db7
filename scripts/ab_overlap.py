#!/usr/bin/env python3
"""Config 3 / reference layout: one stream per step (pyramid -> chain -> tail back to back) against ``overlap=True`` (the pyramid of
batch n + 1 on a second stream beside the chain + keypoint tail of batch n; double-buffered pyramid).  Alternating rounds in one
process, wall clock around K steps + device synchronize (the two internal streams are invisible to events on the current stream);
outputs of the two pipelines compared bit for bit.
    python scripts/ab_overlap.py [workload=config3] [frames] [rounds=6] [steps=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pysilent_amd import distributed as D

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
wl = bench.WORKLOADS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) else wl["frames"]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda", 0)
consts = None
frames = bench.make_frames(torch, D, wl, B, 0, 1, dev)
pipes = {"serial": bench.make_pipeline(wl, B, 0, consts, overlap=False), "overlap": bench.make_pipeline(wl, B, 0, consts, overlap="auto"),
         "overlap, first stream pair unmeasured": bench.make_pipeline(wl, B, 0, consts, overlap="force")}
print("overlap_tuning:", pipes["overlap"].overlap_tuning)


def run(pipe, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.step(frames)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for p in pipes.values():
    run(p, 30)
res = {k: [] for k in pipes}
for rnd in range(rounds):
    for k, p in pipes.items():
        res[k].append(run(p, steps))
for k in pipes:
    print("%-50s ms per %d frames: %s  median %.4f" % (k, B, " ".join("%.4f" % t for t in res[k]), float(np.median(res[k]))))
a, b = pipes["serial"].outputs(allow_truncated=True), pipes["overlap"].outputs(allow_truncated=True)
same = True
for key in ("pyramid", "orient", "line_end", "cs", "end"):
    if key in a:
        eq = torch.equal(a[key].data.view(torch.int32), b[key].data.view(torch.int32))
        same &= eq
        print("%s bit-identical: %s" % (key, eq))
eq = True
if "keypoints" in a:
    eq = np.array_equal(a["keypoint_counts"], b["keypoint_counts"]) and all(np.array_equal(x, y) for x, y in zip(a["keypoints"], b["keypoints"]))
    print("keypoints identical: %s (max rows in a frame %d)" % (eq, int(a["keypoint_counts"].max())))
sys.exit(0 if same and eq else 1)
