#!/usr/bin/env python3
"""config 2: the gray pass of 64 frames as ONE silent_gray_pass_dev call or as calls over chunks of frames (pointer offsets):
does the filter kernel of levels >= 1 find more of the chunk's pyramid in the Infinity Cache?  scripts/ab_chunks.py [chunk ...]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline

B = 64
chunks = [int(a) for a in sys.argv[1:]] or [64, 32, 16, 8, 4]
pipe = LineEndPipeline((1080, 1920), mode="gray", n_levels=5, n_orient=4, batch=B, device=0)
frames = torch.randint(0, 256, (B, 1080, 1920, 1), device="cuda").float()
lib, ctx = pipe._lib, pipe.ctx
K, P, HW = pipe.n_orient, pipe.frame_px, 1080 * 1920

def step(chunk):
    s = pipe._stream()
    for f0 in range(0, B, chunk):
        p = lambda t, stride: C.c_void_p(t.data_ptr() + 4 * stride * f0)
        ctx.check(lib.silent_gray_pass_dev(ctx.handle, pipe.plan.handle, p(frames, HW), chunk,
                                           C.c_void_p(pipe.consts["cs"].ctypes.data), C.c_void_p(pipe.consts["end"].ctypes.data), K,
                                           pipe.clip_hi, p(pipe.pyr, P), p(pipe.cs, P), p(pipe.end, P * K), s))

for _ in range(40):
    step(B)
torch.cuda.synchronize()
res = {c: [] for c in chunks}
for rnd in range(6):
    for c in chunks:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            step(c)
        b.record()
        torch.cuda.synchronize()
        res[c].append(a.elapsed_time(b) / 10)
for c in chunks:
    print("chunk %3d frames: step median %.4f ms  min %.4f" % (c, np.median(res[c]), np.min(res[c])))
