#!/usr/bin/env python3
"""Interleaved A/B timing of gray_line_end_kernel variants in ONE process (dev tool, GPU box).
Variants are selected through the GRAY tuning knob of the context (silent_set_tuning)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline

B = 64
pipe = LineEndPipeline((1080, 1920), mode="gray", n_levels=5, n_orient=4, batch=B, device=0)
frames = torch.randint(0, 256, (B, 1080, 1920, 1), device="cuda").float()
pipe.run_pyramid(frames)
torch.cuda.synchronize()
variants = [v for v in (sys.argv[1:] or ["0", "1"])]
times = {v: [] for v in variants}
for rnd in range(12):
    for v in variants:
        pipe.ctx.set_tuning(0, int(v))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            pipe.run_filters()
        b.record()
        torch.cuda.synchronize()
        if rnd >= 2:
            times[v].append(a.elapsed_time(b) / 5)
byt = pipe.filter_bytes_per_frame() * B
for v in variants:
    t = np.array(times[v])
    print("opts=%s  median %.4f ms  min %.4f  max %.4f   %.0f GB/s (median)" % (v, np.median(t), t.min(), t.max(), byt / np.median(t) / 1e6))
