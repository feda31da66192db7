#!/usr/bin/env python3
"""The product's placement tuner (LineEndPipeline.tune_placement) on a bench workload, one fresh process: prints the record.
    python3 scripts/tune_check.py config5 [key=value tuner arguments]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pysilent_amd import distributed as D
name = sys.argv[1] if len(sys.argv) > 1 else "config5"
kw = {k: float(v) if "." in v else int(v) for k, v in (a.split("=") for a in sys.argv[2:])}
wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
frames = bench.make_frames(torch, D, wl, B, 0, 1, torch.device("cuda", 0))
torch.cuda.synchronize()
rec = pipe.tune_placement(frames, **kw)
print(name, json.dumps(rec), flush=True)
print(name, "settled step after tuning: %.4f ms" % pipe._time_step(frames, 30), flush=True)
