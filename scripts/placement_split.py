#!/usr/bin/env python3
"""Round 6 placement experiment 11: after the product tuner (small maps drawn TOGETHER), does drawing the CS map and the pyramid
SEPARATELY (each behind spacers of its own) find a faster relation still?
    python3 scripts/placement_split.py config5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pysilent_amd import distributed as D
name = sys.argv[1] if len(sys.argv) > 1 else "config5"
wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
frames = bench.make_frames(torch, D, wl, B, 0, 1, torch.device("cuda", 0))
torch.cuda.synchronize()
rec = pipe.tune_placement(frames)
print(name, "tuner:", rec["tries_ms"], "->", rec["chosen_ms"], flush=True)
names = ("cs", "pyr") if wl["mode"] == "gray" else ("orient", "pyr")
held = []
best = pipe._time_step(frames, 20)
print(name, "after the tuner: %.4f" % best, flush=True)
for which in names + names:
    out = []
    for i in range(6):
        held.append(torch.empty(6 << 30, dtype=torch.uint8, device="cuda"))
        old = pipe._pyrs[0] if which == "pyr" else getattr(pipe, which)
        new = torch.empty_like(old)
        if which == "pyr":
            pipe._pyrs[0] = pipe.pyr = new
        else:
            setattr(pipe, which, new)
        for _ in range(4):
            pipe.step(frames)
        t = pipe._time_step(frames, 12)
        out.append(round(t, 4))
        if t < best:
            best = t
            held.append(old)
        else:                       # put the old one back
            held.append(new)
            if which == "pyr":
                pipe._pyrs[0] = pipe.pyr = old
            else:
                setattr(pipe, which, old)
    print(name, "only %-6s drawn again:" % which, out, "best so far %.4f" % best, flush=True)
