#!/usr/bin/env python3
"""Config 3's pyramid walk alone and the whole one-stream step for the library selected by SILENT_LIB_PATH, settled: 40 warm steps, then
medians of 5 windows of 40 launches (pyramid) / 20 steps.  Alternate builds with ROUNDS=2 scripts/ab_walk_libs.sh lib1.so lib2.so ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline

wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
B = 32
kw = dict(center_dimensions=(288, 192), scale=np.e ** .5) if wl == "reference_layout" else {}
pipe = LineEndPipeline((1080, 1920), mode="rgb", n_levels=6, batch=B, device=0, selection=True, value_map=False, peak_value_map=False, **kw)
frames = torch.stack([torch.from_numpy(np.random.default_rng(i).integers(0, 256, (1080, 1920, 3)).astype(np.float32)) for i in range(B)]).cuda()
for _ in range(40):
    pipe.step(frames)
torch.cuda.synchronize()


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


pyr, step = [], []
for _ in range(5):
    pyr.append(timed(lambda: pipe.run_pyramid(frames), 40))
    step.append(timed(lambda: pipe.step(frames), 20))
print("pyramid %.4f  step %.4f" % (float(np.median(pyr)), float(np.median(step))))
