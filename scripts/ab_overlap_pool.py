#!/usr/bin/env python3
"""Does overlap=True depend on WHICH streams of torch's pool a pipeline gets?  For k = 0 .. 6: create (and drop) k overlap pipelines
first, then time the reference layout on one stream and with overlap=True in the (k + 1)-th pipeline.
    python scripts/ab_overlap_pool.py [workload=reference_layout]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pysilent_amd import distributed as D

name = sys.argv[1] if len(sys.argv) > 1 else "reference_layout"
wl = bench.WORKLOADS[name]
B = wl["frames"]
dev = torch.device("cuda", 0)
frames = bench.make_frames(torch, D, wl, B, 0, 1, dev)


def run(pipe, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.step(frames)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


from pysilent_amd.pipeline import _spin_ms
a, b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
print("spin kernel: one stream %.3f ms, two streams %.3f ms" % (_spin_ms(torch, dev, [a], 1 << 20), _spin_ms(torch, dev, [a, b], 1 << 20)))
serial = bench.make_pipeline(wl, B, 0, None, overlap=False)
run(serial, 40)
for k in range(7):
    p = bench.make_pipeline(wl, B, 0, None, overlap=os.environ.get("AB_OVERLAP", "1") == "auto" and "auto" or "force")
    run(p, 40)
    t_o = float(np.median([run(p, 30) for _ in range(3)]))
    t_s = float(np.median([run(serial, 30) for _ in range(3)]))
    print("pipeline #%d with streams: walk stream %#x  chain stream %#x  verified concurrent %s   overlap %.4f  one stream %.4f" % (
        k + 1, p._walk_stream.cuda_stream, p._chain_stream.cuda_stream, getattr(p, "overlap_verified", None), t_o, t_s), flush=True)
    if p.overlap_tuning:
        print("      tuning:", p.overlap_tuning, flush=True)
    del p
    torch.cuda.empty_cache()
