#!/usr/bin/env python3
"""Per-level timing of the pyramid kernel and of the filter kernel (dev tool, GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd import _runtime
from pysilent_amd.util.zoom.from_image import classic_levels
import ctypes as C
from pysilent_amd import _lib

def time_plan(levels, frames, reps=20):
    B, H, W, Cc = frames.shape
    plan = _runtime.PyramidPlan(H, W, Cc, levels, 0)
    out = torch.empty(B * plan.frame_px * Cc, dtype=torch.float32, device='cuda')
    lib = _lib.load(); ctx = plan.ctx
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        ctx.check(lib.silent_pyramid_dev(ctx.handle, plan.handle, C.c_void_p(frames.data_ptr()), B, C.c_void_p(out.data_ptr()), s))
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): run()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, plan.frame_px

if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    H, W = 1080, 1920
    frames = torch.randint(0, 256, (B, H, W, Cc), device='cuda').float()
    allv = classic_levels((H, W), 2.0, G)
    per_level = [] if os.environ.get('PYR_ALL_ONLY') else [('L%d' % i, [allv[i]]) for i in range(G)]
    for name, lv in [('all', allv)] + per_level:
        ms, px = time_plan(lv, frames)
        rd = H * W * 4 * B * Cc; wr = px * 4 * B * Cc
        print('%-4s %.4f ms  out px/frame %8d  write GB/s %.0f  (read+write)/t GB/s %.0f' % (name, ms, px, wr / ms / 1e6, (rd + wr) / ms / 1e6))
