#!/usr/bin/env python3
"""Why does one workload's step time spread more than the others' across (and within) calls?  Runs a workload's step back to back
for ~SECONDS and prints a time series: ms per step over windows of 20 steps next to what rocm-smi reports at that moment (shader /
memory / fabric clocks, package power, edge / junction / HBM temperatures, throttle status if exposed).  If the step time moves
WITH the clock or power inside one run, the box is power / thermally managed for that kernel; if it is flat inside a run and
differs between boxes, it is the device.
    python scripts/step_spread.py [workload=config5] [seconds=10]"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pysilent_amd import distributed as D

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
wl = bench.WORKLOADS[name]
B = wl["frames"]
dev = torch.device("cuda", 0)
pipe = bench.make_pipeline(wl, B, 0, None)
frames = bench.make_frames(torch, D, wl, B, 0, 1, dev)


def smi():
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        d = json.loads(r.stdout)
        c = d.get("card0", next(iter(d.values())))
    except (ValueError, StopIteration):
        return {}
    keep = {}
    for k, v in c.items():
        kl = k.lower()
        if any(s in kl for s in ("sclk", "mclk", "fclk", "power", "temperature")):
            keep[k.replace("Temperature (Sensor ", "T(").replace(" (C)", "").replace("clock speed:", "").strip()] = v
    return keep


torch.cuda.synchronize()
print("idle:", smi(), flush=True)
t0 = time.time()
series = []
while time.time() - t0 < seconds:
    ws = []
    for _ in range(5):                    # five windows of 20 steps, then one sample with 60 more steps queued
        torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(20):
            pipe.step(frames)
        torch.cuda.synchronize()
        ws.append((time.perf_counter() - a) / 20 * 1e3)
    for _ in range(60):
        pipe.step(frames)
    s = smi()
    torch.cuda.synchronize()
    series.append((time.time() - t0, ws, s))
    print("t %5.1f s  ms/step %s  %s" % (series[-1][0], " ".join("%.3f" % w for w in ws), s), flush=True)
allw = np.array([w for _, ws, _ in series for w in ws])
print("%s: %d windows, ms/step min %.3f  median %.3f  max %.3f  (max / min %.2f)" % (name, allw.size, allw.min(), np.median(allw), allw.max(), allw.max() / allw.min()))
