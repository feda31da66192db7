#!/usr/bin/env python3
"""Per-frame latency of the reference application graph (LineEndDisplayer.callback) on one 640x480 RGB frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.recognition_testing import LineEndDisplayer
disp = LineEndDisplayer(use_graph="--graph" in sys.argv)
print("HIP graph replay" if disp.use_graph else "eager launches")
frame = np.random.default_rng(0).integers(0, 256, (480, 640, 3)).astype(np.uint8)
for _ in range(10):
    disp.callback(frame)
torch.cuda.synchronize()
ts = []
for _ in range(100):
    t0 = time.perf_counter(); disp.callback(frame); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("callback: median %.3f ms  p90 %.3f  min %.3f  (%.0f frames/s)" % (np.median(ts), np.percentile(ts, 90), ts.min(), 1e3 / np.median(ts)))
from pysilent_amd.util import zoom
z = zoom.from_image(frame.astype(np.float32), 3, disp.output_size, disp.zoom_ratio)
ts = []
for _ in range(100):
    t0 = time.perf_counter(); disp.run_device(z); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("run_device (graph only, host pyramid in): median %.3f ms" % (np.median(ts) * 1e3))
ts = []
for _ in range(100):
    t0 = time.perf_counter(); zoom.from_image(frame.astype(np.float32), 3, disp.output_size, disp.zoom_ratio); ts.append(time.perf_counter() - t0)
print("zoom.from_image (host in, host out): median %.3f ms" % (np.median(ts) * 1e3))
