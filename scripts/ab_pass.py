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
variants = {"two-step": ("0", two_step), "walk": (os.environ.get("SILENT_AB_WALK_OPTS", "128"), lambda: pipe.step(frames)),
            "walk plain st": ("384", lambda: pipe.step(frames)), "walk+region": ("640", lambda: pipe.step(frames)),
            "walk+region 3/CU": (str(640 + 1024), lambda: pipe.step(frames)), "walk+region 4/CU": (str(640 + 2048), lambda: pipe.step(frames)),
            "walk 3/CU": ("1152", lambda: pipe.step(frames)),
            "walk1": (str(1 << 18), lambda: pipe.step(frames)),
            "A seg64": (str(640 + (2 << 12)), lambda: pipe.step(frames)), "A seg128": (str(640 + (4 << 12)), lambda: pipe.step(frames)),
            "A seg272": (str(640 + (9 << 12)), lambda: pipe.step(frames)), "A seg544": (str(640 + (17 << 12)), lambda: pipe.step(frames)),
            "A seg1088": (str(640 + (34 << 12)), lambda: pipe.step(frames)),
            "tile stream": ("0", lambda: pipe.step(frames))}
if os.environ.get("AB_ONLY"):
    variants = {k: v for k, v in variants.items() if k in os.environ["AB_ONLY"].split(",")}
if os.environ.get("AB_QUICK"):
    variants = {"stream": variants["stream"]}
times = {k: [] for k in variants}
for rnd in range(12):
    for k, (opt, fn) in variants.items():
        pipe.ctx.set_tuning(0, int(opt))
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
    print("%-18s median %.4f ms  min %.4f  max %.4f   %.0f GB/s algorithmic = %.1f %% of 8 TB/s" % (k, np.median(t), t.min(), t.max(), byt / np.median(t) / 1e6, byt / np.median(t) / 1e6 / 80))

# dominant kernel alone (HIP events recorded by the library around its launch) + a device copy for calibration
pipe.ctx.set_tuning(0, int(os.environ.get("AB_BASE_OPTS", "0")))
ks = []
for _ in range(30):
    pipe.set_profiling(1)          # resets the sample ring: one pair per step here
    pipe.step(frames)
    torch.cuda.synchronize()
    ks.append(pipe.profiled_kernel()[0])
pipe.set_profiling(False)
ks = np.array(ks[5:])
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
b_ = torch.empty_like(a)
cs = []
for _ in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b_.copy_(a); e1.record(); torch.cuda.synchronize()
    cs.append(e0.elapsed_time(e1))
copy_gbs = 2 * a.numel() * 4 / (np.median(cs[2:]) * 1e6)
print("stream kernel alone: median %.4f ms  min %.4f   | 1 GiB device copy: %.0f GB/s (read+write)  | kernel_ms x copy_TB/s = %.3f"
      % (np.median(ks), ks.min(), copy_gbs, np.median(ks) * copy_gbs / 1000))
