#!/usr/bin/env python3
"""RGB chain kernel time against the number of 1080p frames in the launch (tile-height threshold, dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd import _runtime as rt
from pysilent_amd.pipeline import default_constants
k = default_constants("rgb")
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 16]:
    x = torch.rand((B, 1080, 1920, 3), device="cuda") * 255
    for _ in range(3): rt.rgb_line_end(x, k)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rt.rgb_line_end(x, k); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("B=%2d  %.3f ms" % (B, float(np.median(ts))))
