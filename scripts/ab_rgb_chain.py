#!/usr/bin/env python3
"""RGB chain kernel A/B on config 3 (32 x 1080p RGB, 6 levels): TUNE_RGB knob values given on the command line
(default: 0 = pair kernel, 16 = one pixel per lane), alternating rounds in one process.
    python scripts/ab_rgb_chain.py [batch] [knob ...]        knob: integer, e.g. 0 16 $((45<<8)) $((16|45<<8))"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
knobs = [int(a, 0) for a in sys.argv[2:]] or [0, 16]
pipe = LineEndPipeline((1080, 1920), mode="rgb", n_levels=6, batch=B, device=0, max_keypoints_per_frame=1 << 16, selection=True)
frames = torch.randint(0, 256, (B, 1080, 1920, 3), device="cuda").float()
pipe.run_pyramid(frames)
torch.cuda.synchronize()
times = {k: [] for k in knobs}
for rnd in range(7):
    for k in knobs:
        pipe.ctx.set_tuning(1, k)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            pipe.run_filters()
        b.record()
        torch.cuda.synchronize()
        if rnd >= 1:
            times[k].append(a.elapsed_time(b) / 5)
pipe.ctx.set_tuning(1, 0)
px = B * sum((1080 >> l) * (1920 >> l) for l in range(6))
for k in knobs:
    m = float(np.median(times[k]))
    print("TUNE_RGB %-6d (th %s, %s) median %.4f ms  min %.4f   %.2f TB/s of 40 B/px" % (
        k, ((k >> 8) & 255) * 2 or "auto", "one px/lane" if k & 16 else "pair", m, np.min(times[k]), px * 40 / m / 1e9))
