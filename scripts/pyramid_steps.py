"""RGB pyramids at zoom steps below the reference's e ** .5: the single-read walk (pyramid_walk3_kernel, 28 / 24 pixels per wave)
against the unit + region kernels (PYRAMID knob 2), same plan, same buffers, alternating.  MI355X:  python scripts/pyramid_steps.py"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pysilent_amd import _runtime as rt  # noqa: E402
from pysilent_amd._lib import TUNE_PYRAMID  # noqa: E402
from pysilent_amd.util.zoom.from_image import classic_levels  # noqa: E402

H, W, B = 1080, 1920, 32


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, scale, n, W in [("e^.5, 6 levels", np.e ** .5, 6, 1920), ("sqrt 2, 8 levels", 2 ** .5, 8, 1920),
                          ("2^(1/3), 8 levels", 2 ** (1 / 3), 8, 1920), ("e^.5, 6 levels, 1918 wide", np.e ** .5, 6, 1918),
                          ("2, 6 levels, 1918 wide", 2.0, 6, 1918), ("2, 6 levels", 2.0, 6, 1920)]:
    frames = torch.randint(0, 256, (B, H, W, 3), device="cuda").float()
    plan = rt.PyramidPlan(H, W, 3, classic_levels((H, W), scale, n))
    run = lambda: plan.run(frames)  # noqa: E731  (the output comes from torch's caching allocator: the same block every call)
    px_out = sum(l[6] * l[7] for l in classic_levels((H, W), scale, n))
    alg = B * (H * W + px_out) * 12
    rows = []
    for rnd in range(3):
        a = timed(run)
        with rt.tuning(TUNE_PYRAMID, 2):
            b = timed(run)
        rows.append((a, b))
    a, b = min(r[0] for r in rows), min(r[1] for r in rows)
    print("%-28s walk plans %-8s  walk %.3f ms (%.2f TB/s)   unit + region %.3f ms (%.2f TB/s)   x%.2f" %
          (name, plan.walk_plans, a, alg / a / 1e9, b, alg / b / 1e9, b / a), flush=True)
