#!/usr/bin/env python3
"""Round 6 placement experiment 10: after the product tuner has placed the maps, does moving the FRAMES (the caller's resident batch)
still change the step?  Eight more allocations of the batch (behind spacers, all kept), the step timed on each.
    python3 scripts/placement_frames.py config2"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pysilent_amd import distributed as D
name = sys.argv[1] if len(sys.argv) > 1 else "config2"
wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
frames = bench.make_frames(torch, D, wl, B, 0, 1, torch.device("cuda", 0))
torch.cuda.synchronize()
rec = pipe.tune_placement(frames)
print(name, "maps:", rec["tries_ms"], "->", rec["chosen_ms"], flush=True)
base = pipe._time_step(frames, 20)
held, out = [frames], [round(base, 4)]
for i in range(8):
    held.append(torch.empty(4 << 30, dtype=torch.uint8, device="cuda"))
    f2 = torch.empty_like(frames)
    f2.copy_(frames)
    held.append(f2)
    for _ in range(5):
        pipe.step(f2)
    out.append(round(pipe._time_step(f2, 20), 4))
print(name, "frames moved (first = where the tuner saw them):", out, flush=True)
