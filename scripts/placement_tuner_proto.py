#!/usr/bin/env python3
"""Round 6 placement experiment 9 (prototype of the informed tuner): the big map stays; only the SMALL maps are drawn again, with
physical-only spacers (hipMemCreate without a mapping) of at most the maps' own size between the draws.  Prints what each part costs.
    python3 scripts/placement_tuner_proto.py config5 [draws] [spacer GiB] [mapped spacers: 0/1]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
n_draws = int(sys.argv[2]) if len(sys.argv) > 2 else 10
spacer_gib = float(sys.argv[3]) if len(sys.argv) > 3 else 7.0
mapped = int(sys.argv[4]) if len(sys.argv) > 4 else 0
probe = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libvmm_probe.so"))
probe.vmm_phys_create.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_ulonglong)]
probe.vmm_phys_release.argtypes = [C.c_ulonglong]

wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
gray = wl["mode"] == "gray"
c = 1 if gray else 3
frames = torch.randint(0, 256, (B,) + wl["hw"] + (c,), device="cuda").float()
small = ("pyr", "cs") if gray else ("pyr", "orient")


def step_ms(n=10, warm=4):
    for _ in range(warm):
        pipe.step(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.step(frames)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def get(k):
    return pipe._pyrs[0] if k == "pyr" else getattr(pipe, k)


def put(k, t):
    if k == "pyr":
        pipe._pyrs[0] = pipe.pyr = t
    else:
        setattr(pipe, k, t)


for _ in range(30):
    pipe.step(frames)
t_all = time.perf_counter()
rec = {"workload": name, "spacer_GiB": spacer_gib, "mapped_spacers": mapped, "draws": []}
first = step_ms()
print("first draw: step %.4f ms" % first, flush=True)
own = {k: get(k) for k in small}
held, handles = [], []
best = (first, dict(own))
for i in range(n_draws):
    t0 = time.perf_counter()
    if spacer_gib:
        if mapped:
            held.append(torch.empty(int(spacer_gib * 2 ** 30), dtype=torch.uint8, device="cuda"))
        else:
            h = C.c_ulonglong(0)
            rc = probe.vmm_phys_create(0, int(spacer_gib * 2 ** 30), C.byref(h))
            if rc:
                sys.exit("vmm_phys_create failed %d" % rc)
            handles.append(h.value)
    t1 = time.perf_counter()
    cand = {k: torch.empty_like(own[k]) for k in small}
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for k in small:
        put(k, cand[k])
    ms = step_ms()
    t3 = time.perf_counter()
    held.append(cand)
    if ms < best[0]:
        best = (ms, cand)
    rec["draws"].append({"step_ms": round(ms, 4), "spacer_s": round(t1 - t0, 4), "alloc_s": round(t2 - t1, 4), "timing_s": round(t3 - t2, 4)})
    print("draw %2d: step %.4f ms   spacer %.3f s  alloc %.3f s  timing %.3f s" % (i + 1, ms, t1 - t0, t2 - t1, t3 - t2), flush=True)
for k in small:
    put(k, best[1][k])
t0 = time.perf_counter()
for h in handles:
    probe.vmm_phys_release(h)
held = None
torch.cuda.synchronize()
print("release %.3f s; total %.2f s; first %.4f -> chosen %.4f ms (check: %.4f)" % (time.perf_counter() - t0, time.perf_counter() - t_all, first, best[0], step_ms()), flush=True)
print(json.dumps(rec), flush=True)
