#!/usr/bin/env python3
"""Round 6 placement experiment 7: is it WHERE the big map lies, or where it lies RELATIVE to the small maps?  The K map is drawn a few
times (all kept, 8 GiB apart); then, for the first and for the last draw of it, the CS map and the pyramid are re-allocated far
away (16 GiB spacers) several times.  Real kernel on every combination.
    python3 scripts/placement_pairs.py config5"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
big_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
small_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 16.0
n_small = int(sys.argv[4]) if len(sys.argv) > 4 else 5
wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
frames = torch.randint(0, 256, (B,) + wl["hw"] + (1,), device="cuda").float()


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


def gib(n):
    return torch.empty(int(n * 2 ** 30), dtype=torch.uint8, device="cuda")


for _ in range(30):
    pipe.step(frames)
ends = [pipe.end]
small = [(pipe.cs, pipe._pyrs[0])]
held = []
for i in range(5):
    if big_gap:
        held.append(gib(big_gap))
    ends.append(torch.empty_like(ends[0]))
for i in range(n_small):
    if small_gap:
        held.append(gib(small_gap))
    small.append((torch.empty_like(small[0][0]), torch.empty_like(small[0][1])))
print("rows: K-map draw (%g GiB spacers); columns: CS + pyramid draw (%g GiB spacers, the first one is the pipeline's own, allocated beside K-map draw 0)" % (big_gap, small_gap))
table = []
for ei, e in enumerate(ends):
    pipe.end = e
    row = []
    for (cs, pyr) in small:
        pipe.cs = cs
        pipe._pyrs[0] = pipe.pyr = pyr
        row.append(round(kernel_ms(6, 8), 4))
    table.append(row)
    print("K map %d:  " % ei + "  ".join("%.4f" % v for v in row), flush=True)
# and the frames: re-allocated far away, on the first K map and the pipeline's own small maps
pipe.end, pipe.cs = ends[0], small[0][0]
pipe._pyrs[0] = pipe.pyr = small[0][1]
fr = []
for i in range(4):
    held.append(gib(16))
    f2 = torch.empty_like(frames)
    f2.copy_(frames)
    held.append(frames)
    frames = f2
    fr.append(round(kernel_ms(6, 8), 4))
print("frames re-allocated 16 GiB further each time (K map 0, own small maps): " + "  ".join("%.4f" % v for v in fr), flush=True)
print(json.dumps({"workload": name, "table": table, "frames_moved": fr}), flush=True)
