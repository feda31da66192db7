#!/usr/bin/env python3
"""One pipeline, one set of buffers, a tuning knob of the library switched on and off between timing windows.
    python scripts/ab_knob.py <workload of bench.py> <gray|rgb|pyramid> <value>   [ROUNDS=5] [STEPS=30]
e.g. reference_layout pyramid 8 = the border pixels of the union walk plans as a launch of their own (silent_pyramid_api.hip)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pysilent_amd import _lib, _runtime  # noqa: E402

name, which, value = sys.argv[1], sys.argv[2], int(sys.argv[3])
knob = {"gray": _lib.TUNE_GRAY, "rgb": _lib.TUNE_RGB, "pyramid": _lib.TUNE_PYRAMID}[which]
rounds, steps = int(os.environ.get("ROUNDS", "5")), int(os.environ.get("STEPS", "30"))
wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
ch = 1 if wl["mode"] == "gray" else 3
frames = torch.randint(0, 256, (B,) + wl["hw"] + (ch,), device="cuda").float()


def timed(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        pipe.step(frames)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ctx = _runtime.get_context()
for _ in range(20):
    pipe.step(frames)
res = {0: [], value: []}
for r in range(rounds):
    for v in (0, value):
        ctx.set_tuning(knob, v)
        timed(5)
        res[v].append(timed(steps))
ctx.set_tuning(knob, 0)
for v, t in res.items():
    print("%s knob %s = %-4d  step %.4f ms (min %.4f)" % (name, which, v, np.median(t), np.min(t)), flush=True)
