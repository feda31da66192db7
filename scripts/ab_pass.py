#!/usr/bin/env python3
"""Interleaved A/B timing of the whole gray pass (silent_gray_pass_dev) and of the two-step path, one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline

B = 64
pipe = LineEndPipeline((1080, 1920), mode="gray", n_levels=5, n_orient=4, batch=B, device=0)
frames = torch.randint(0, 256, (B, 1080, 1920, 1), device="cuda").float()
def two_step():
    pipe.run_pyramid(frames); pipe.run_filters()
variants = {"two-step": ("0", two_step), "stream": ("0", lambda: pipe.step(frames)),
            "stream+xcd": ("32", lambda: pipe.step(frames)), "no-stream": ("16", lambda: pipe.step(frames))}
times = {k: [] for k in variants}
for rnd in range(12):
    for k, (opt, fn) in variants.items():
        os.environ["SILENT_GRAY_OPTS"] = opt
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn()
        b.record()
        torch.cuda.synchronize()
        if rnd >= 2:
            times[k].append(a.elapsed_time(b) / 5)
byt = pipe.algorithmic_bytes_per_frame() * B
for k in variants:
    t = np.array(times[k])
    print("%-10s median %.4f ms  min %.4f  max %.4f   %.0f GB/s algorithmic = %.1f %% of 8 TB/s" % (k, np.median(t), t.min(), t.max(), byt / np.median(t) / 1e6, byt / np.median(t) / 1e6 / 80))
