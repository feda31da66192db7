#!/usr/bin/env python3
"""What shader clock and power does the GPU report while a kernel of this library runs back to back?
(rocm-smi sampled from a second process during ~6 s of launches; idle before and after.)
    python scripts/clock_under_load.py [chain|pyramid|gray]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline

what = sys.argv[1] if len(sys.argv) > 1 else "chain"


def smi(tag):
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showuse"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    keep = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "GPU use"))]
    print(tag, " | ".join(k.split("GPU[0]")[-1].strip(" :\t") for k in keep if "GPU[0]" in k), flush=True)


if what == "gray":
    B = 64
    pipe = LineEndPipeline((1080, 1920), mode="gray", n_levels=5, batch=B, device=0)
    frames = torch.rand((B, 1080, 1920, 1), device="cuda") * 255
    fn = lambda: pipe.step(frames)
else:
    B = 32
    pipe = LineEndPipeline((1080, 1920), mode="rgb", n_levels=6, batch=B, device=0, max_keypoints_per_frame=1 << 16, selection=True,
                           value_map=False, peak_value_map=False)
    frames = torch.randint(0, 256, (B, 1080, 1920, 3), device="cuda").float()
    pipe.step(frames)
    fn = pipe.run_filters_keypoints if what == "chain" else (lambda: pipe.run_pyramid(frames))
torch.cuda.synchronize()
smi("idle  ")
t0 = time.time()
k = 0
while time.time() - t0 < 6.0:
    for _ in range(200):
        fn()
    torch.cuda.synchronize()      # (keeps the queue bounded; the gap is microseconds)
    k += 1
    if k % 4 == 0:
        for _ in range(400):
            fn()
        smi("load  ")             # sampled while 400 launches are queued
        torch.cuda.synchronize()
smi("after ")
