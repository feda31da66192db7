#!/usr/bin/env python3
"""Config 3: the chain kernel's extrema instantiation (silent_rgb_keypoints: 3 waves / SIMD) against the plain one
(silent_rgb_line_end: 4 waves / SIMD) on the same pyramid; alternating rounds of 10."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline

B = 32
frames = torch.stack([torch.from_numpy(np.random.default_rng(i).integers(0, 256, (1080, 1920, 3)).astype(np.float32)) for i in range(B)]).cuda()
kp = LineEndPipeline((1080, 1920), mode="rgb", n_levels=6, batch=B, device=0, max_keypoints_per_frame=1 << 16, selection=True,
                     value_map=False, peak_value_map=False)
pl = LineEndPipeline((1080, 1920), mode="rgb", n_levels=6, batch=B, device=0, selection=False, value_map=True)
pl.pyr = kp.pyr
for _ in range(10):
    kp.step(frames)
    pl.run_filters()
torch.cuda.synchronize()


def timed(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {"keypoints (chain MM + tail)": [], "line_end (chain plain)": []}
for rnd in range(6):
    res["keypoints (chain MM + tail)"].append(timed(kp.run_filters_keypoints))
    res["line_end (chain plain)"].append(timed(pl.run_filters))
for k, v in res.items():
    print("%-30s %.4f ms" % (k, float(np.median(v[1:]))))
