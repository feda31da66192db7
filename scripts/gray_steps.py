"""Gray whole passes (pyramid + CS + 4-orientation line-end, 64 x 1080p) at zoom ratios below the reference's e ** .5: the single-read
stream kernel (dense slot layout, gray_stream_kernel<K, 7, 1>) against the unit-fused + region path (GRAY knob 16), same plan, same
buffers, alternating.  MI355X:  python scripts/gray_steps.py"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pysilent_amd import _runtime as rt  # noqa: E402
from pysilent_amd._lib import TUNE_GRAY  # noqa: E402
from pysilent_amd.pipeline import default_constants  # noqa: E402
from pysilent_amd.util.zoom.from_image import classic_levels  # noqa: E402

H, W, B = 1080, 1920, 64
frames = torch.randint(0, 256, (B, H, W, 1), device="cuda").float()


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


consts = default_constants("gray", 4)
cs_k, end_k = consts["cs"], consts["end"]
for name, scale, n in [("2, 5 levels", 2.0, 5), ("e^.5, 6 levels", np.e ** .5, 6), ("sqrt 2, 8 levels", 2 ** .5, 8), ("1.5, 6 levels", 1.5, 6)]:
    levels = classic_levels((H, W), scale, n)
    plan = rt.PyramidPlan(H, W, 1, levels)
    px = sum(l[6] * l[7] for l in levels)
    alg = B * (H * W * 4 + px * 4 * 6)      # frame in; pyramid + cs + 4 orientations out
    run = lambda: plan.gray_pass(frames, cs_k, end_k)  # noqa: E731
    rows = []
    for _ in range(3):
        a = timed(run)
        with rt.tuning(TUNE_GRAY, 16):
            b = timed(run)
        rows.append((a, b))
    a, b = min(r[0] for r in rows), min(r[1] for r in rows)
    print("%-18s streamable %-5s  stream %.3f ms (%.2f TB/s)   unit-fused + region %.3f ms (%.2f TB/s)   x%.2f" %
          (name, plan.streamable, a, alg / a / 1e9, b, alg / b / 1e9, b / a), flush=True)
