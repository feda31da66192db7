#!/usr/bin/env python3
"""Config 3 (32 x 1080p RGB, 6 levels), one stream: chain kernel alone (HIP events the library records around its launch),
chain + keypoint tail, whole step, under alternating SILENT_TUNE_RGB knob values in one process (rounds of 10 steps, medians of 6
rounds after one discarded).  Usage: ab_rgb_knobs2.py 0 128 [...]   (128: 16-byte stores instead of the default 12-byte ones)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline
from pysilent_amd._lib import TUNE_RGB

knobs = [int(a, 0) for a in sys.argv[1:]] or [0, 128]
B = 32
pipe = LineEndPipeline((1080, 1920), mode="rgb", n_levels=6, batch=B, device=0, selection=True, value_map=False, peak_value_map=False)
frames = torch.stack([torch.from_numpy(np.random.default_rng(i).integers(0, 256, (1080, 1920, 3)).astype(np.float32)) for i in range(B)]).cuda()
for _ in range(30):
    pipe.step(frames)
torch.cuda.synchronize()


def timed(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def kernel_ms():
    pipe.set_profiling(1)
    for _ in range(8):
        pipe.run_filters_keypoints()
    torch.cuda.synchronize()
    t, _ = pipe.profiled_kernel()
    pipe.set_profiling(0)
    return t


res = {k: {"chain kernel": [], "chain+tail": [], "step": []} for k in knobs}
for rnd in range(7):
    for k in knobs:
        pipe.ctx.set_tuning(TUNE_RGB, k)
        pipe.step(frames)
        ck, c, s = kernel_ms(), timed(pipe.run_filters_keypoints), timed(lambda: pipe.step(frames))
        if rnd:
            res[k]["chain kernel"].append(ck)
            res[k]["chain+tail"].append(c)
            res[k]["step"].append(s)
pipe.ctx.set_tuning(TUNE_RGB, 0)
for k in knobs:
    print("knob %-4d  " % k + "  ".join("%s %.4f" % (n, float(np.median(v))) for n, v in res[k].items()))
