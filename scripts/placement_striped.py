#!/usr/bin/env python3
"""Round 6 placement experiment 8: maps whose physical chunks are CO-LOCATED by construction (scripts/exp/vmm_probe.hip: vmm_striped:
the chunks backing fraction t of every map are created next to each other) against plain allocations, alternating, all kept,
with spacers between the draws so that they land in different stretches of the device memory.
    python3 scripts/placement_striped.py config5 [draws] [chunk MiB] [spacer GiB] [frames too: 0/1]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
n_draws = int(sys.argv[2]) if len(sys.argv) > 2 else 6
chunk_mib = int(sys.argv[3]) if len(sys.argv) > 3 else 32
spacer = float(sys.argv[4]) if len(sys.argv) > 4 else 6.0
with_frames = int(sys.argv[5]) if len(sys.argv) > 5 else 1
probe = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libvmm_probe.so"))
probe.vmm_striped.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(C.c_void_p)]

wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
gray = wl["mode"] == "gray"
c = 1 if gray else 3
frames0 = torch.randint(0, 256, (B,) + wl["hw"] + (c,), device="cuda").float()
frames = frames0
names = ("pyr", "cs", "end") if gray else ("pyr", "orient", "line_end")


class Raw(object):
    def __init__(self, ptr, like):
        self.ptr, self.like = int(ptr), like

    def data_ptr(self):
        return self.ptr


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


def step_ms(n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        pipe.step(frames)
    a.record()
    for _ in range(n):
        pipe.step(frames)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def adopt(maps):
    pipe._pyrs[0] = pipe.pyr = maps[0]
    setattr(pipe, names[1], maps[1])
    setattr(pipe, names[2], maps[2])


for _ in range(30):
    pipe.step(frames)
own = [pipe._pyrs[0], getattr(pipe, names[1]), getattr(pipe, names[2])]
held = []
rec = {"workload": name, "chunk_MiB": chunk_mib, "spacer_GiB": spacer, "plain": [], "striped": []}
for i in range(n_draws):
    if i:
        if spacer:
            held.append(torch.empty(int(spacer * 2 ** 30), dtype=torch.uint8, device="cuda"))
        maps = [torch.empty_like(t) for t in own]
        fr = torch.empty_like(frames0)
        fr.copy_(frames0)
    else:
        maps, fr = own, frames0
    held.append((maps, fr))
    adopt(maps)
    frames = fr
    rec["plain"].append([round(kernel_ms(), 4), round(step_ms(), 4)])
    print("plain   draw %d: kernel %.4f ms  step %.4f ms" % (i, *rec["plain"][-1]), flush=True)
    sizes = [t.numel() * 4 for t in own] + ([frames0.numel() * 4] if with_frames else [])
    arr = (C.c_size_t * len(sizes))(*sizes)
    out = (C.c_void_p * len(sizes))()
    rc = probe.vmm_striped(0, len(sizes), arr, chunk_mib << 20, out)
    if rc:
        sys.exit("vmm_striped failed: %d" % rc)
    adopt([Raw(out[k], own[k]) for k in range(3)])
    if with_frames:
        # a torch view of the striped frame buffer (the pipeline checks the frames' shape): copy the frames there
        import numpy as np

        class _Arr(object):
            pass

        holder = _Arr()
        holder.__cuda_array_interface__ = {"shape": tuple(frames0.shape), "typestr": "<f4", "data": (int(out[3]), False), "version": 2}
        fr2 = torch.as_tensor(holder, device="cuda")
        fr2.copy_(frames0)
        frames = fr2
    else:
        frames = frames0
    rec["striped"].append([round(kernel_ms(), 4), round(step_ms(), 4)])
    print("striped draw %d: kernel %.4f ms  step %.4f ms" % (i, *rec["striped"][-1]), flush=True)
print(json.dumps(rec), flush=True)
