#!/usr/bin/env python3
"""rgb_line_end_kernel time against the tile height (config 3: 32 x 1080p RGB, 6 levels), one process, alternating rounds.
The library's own choice (cost model: rounds x (th + 14)) is the "auto" row."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pipe = LineEndPipeline((1080, 1920), mode="rgb", n_levels=6, batch=B, device=0, max_keypoints_per_frame=1 << 16, selection=True,
                       value_map=False, peak_value_map=False)
run = pipe.run_filters_keypoints if os.environ.get("SWEEP_KP") else pipe.run_filters   # (the extrema instantiation + sparse tail | the plain chain)
frames = torch.randint(0, 256, (B, 1080, 1920, 3), device="cuda").float()
pipe.run_pyramid(frames)
torch.cuda.synchronize()
ths = [int(t) for t in os.environ.get("SWEEP_TH", "0,50,72,90,96,100,104,108,120,136,156").split(",")]
times = {t: [] for t in ths}
for rnd in range(6):
    for t in ths:
        pipe.ctx.set_tuning(1, (t // 2) << 8)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            run()
        b.record()
        torch.cuda.synchronize()
        if rnd >= 1:
            times[t].append(a.elapsed_time(b) / 5)
pipe.ctx.set_tuning(1, 0)
for t in ths:
    print("th %-5s median %.4f ms  min %.4f" % ("auto" if t == 0 else t, np.median(times[t]), np.min(times[t])))
