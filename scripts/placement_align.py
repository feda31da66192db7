#!/usr/bin/env python3
"""Round 6 store-pattern experiment: does the 1-channel maps' cost come from their runs not being whole 64-byte pieces?  The
synthetic three-stream kernel with 56 output columns per wave (the kernel's: 224-byte runs = 3.5 x 64 B) against 64 columns (256-byte
runs, 64-byte aligned), on three draws of the big map.
    python3 scripts/placement_align.py config5|config2"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
probe = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libvmm_probe.so"))
probe.wp2_run.restype = C.c_float
probe.wp2_run.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_longlong, C.c_int, C.c_int]
wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
frames = torch.randint(0, 256, (B,) + wl["hw"] + (1,), device="cuda").float()
h, w = wl["hw"]
K = wl["n_orient"]


def kernel_ms(warm=10, timed=8):
    for _ in range(warm):
        pipe.step(frames)
    pipe.set_profiling(1)
    for _ in range(timed):
        pipe.step(frames)
    torch.cuda.synchronize()
    t = pipe.profiled_kernel()[0]
    pipe.set_profiling(0)
    return t


def synth(buf, flags):
    return float(probe.wp2_run(C.c_void_p(buf.data_ptr()), C.c_void_p(pipe.cs.data_ptr()), C.c_void_p(pipe._pyrs[0].data_ptr()),
                               C.c_void_p(frames.data_ptr()), K, w, h, B, pipe.frame_px, flags, 6))


for _ in range(30):
    pipe.step(frames)
held = [pipe.end]
for i in range(4):
    if i:
        held.append(torch.empty(8 << 30, dtype=torch.uint8, device="cuda"))
        held.append(torch.empty_like(held[0]))
        pipe.end = held[-1]
    buf = pipe.end
    print("draw %d: real %.4f | K alone: 56 cols %.4f  64 cols %.4f  48 cols %.4f | 3 store streams: 56 cols %.4f  64 cols %.4f  48 cols %.4f | + frame reads: 56 cols %.4f  64 cols %.4f  48 cols %.4f"
          % (i, kernel_ms(), synth(buf, 0), synth(buf, 128), synth(buf, 256), synth(buf, 6), synth(buf, 6 | 128), synth(buf, 6 | 256),
             synth(buf, 7), synth(buf, 7 | 128), synth(buf, 7 | 256)), flush=True)
