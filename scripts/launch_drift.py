#!/usr/bin/env python3
"""Why does gray_stream_kernel's launch time drift upward over consecutive launches (843 -> 1005 us in the round-1 traces)?
Times every launch of the dominant kernel (HIP events recorded by the library around it) in three regimes:
  A  back-to-back steps, no host gaps              (what the bench and the rocprofv3 traces do)
  B  the same with a 20 ms idle gap before every step   (lets the chip's power management recover between launches)
  C  back-to-back again, after B                    (is the slow state re-entered?)
and, beside each regime, a 1 GiB device copy (pure HBM streaming, no library code) as the control.
If B is flat at the fast value and the copy slows down in A / C by the same factor as the kernel, the drift is the
chip lowering its clocks under sustained memory load (DVFS), not a property of the kernel (cache state, allocator, ...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline

B = 64
pipe = LineEndPipeline((1080, 1920), mode="gray", n_levels=5, n_orient=4, batch=B, device=0)
frames = torch.randint(0, 256, (B, 1080, 1920, 1), device="cuda").float()
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
b = torch.empty_like(a)


def kernel_series(n, gap_s):
    out = []
    for _ in range(n):
        if gap_s:
            torch.cuda.synchronize()
            time.sleep(gap_s)
        pipe.set_profiling(1)
        pipe.step(frames)
        torch.cuda.synchronize()
        out.append(pipe.profiled_kernel()[0] * 1e3)
    pipe.set_profiling(0)
    return np.array(out)


def burst(n):
    """n back-to-back steps with NO synchronisation in between, the kernel timed on the first and the last 8 of them"""
    pipe.set_profiling(1)
    for _ in range(8):
        pipe.step(frames)
    torch.cuda.synchronize()
    first = pipe.profiled_kernel()[0] * 1e3
    pipe.set_profiling(0)
    for _ in range(n - 16):
        pipe.step(frames)
    pipe.set_profiling(1)
    for _ in range(8):
        pipe.step(frames)
    torch.cuda.synchronize()
    last = pipe.profiled_kernel()[0] * 1e3
    pipe.set_profiling(0)
    return first, last


def copy_us(gap_s=0.0, n=6):
    ts = []
    for _ in range(n):
        if gap_s:
            torch.cuda.synchronize()
            time.sleep(gap_s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


for _ in range(5):
    pipe.step(frames)
torch.cuda.synchronize()
time.sleep(0.5)
fmt = lambda s: " ".join("%4.0f" % v for v in s)
sA = kernel_series(40, 0.0)
print("A back-to-back (sync per launch)   us:", fmt(sA[:20]), "...", fmt(sA[-5:]), "| 1 GiB copy %.0f us" % copy_us())
sB = kernel_series(40, 0.02)
print("B 20 ms idle before each launch    us:", fmt(sB[:20]), "...", fmt(sB[-5:]), "| 1 GiB copy %.0f us (20 ms gaps)" % copy_us(0.02))
sC = kernel_series(40, 0.0)
print("C back-to-back again               us:", fmt(sC[:20]), "...", fmt(sC[-5:]), "| 1 GiB copy %.0f us" % copy_us())
time.sleep(0.5)
f, l = burst(400)
print("D 400 steps without any synchronisation: first 8 launches %.0f us, last 8 launches %.0f us" % (f, l))
print("medians  A %.0f  B %.0f  C %.0f  | first 3 of A %.0f  last 10 of A %.0f" % (
    np.median(sA), np.median(sB), np.median(sC), sA[:3].mean(), sA[-10:].mean()))
