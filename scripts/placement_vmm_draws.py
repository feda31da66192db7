#!/usr/bin/env python3
"""Round 6 placement experiment 4: plain allocations (torch.empty -> hipMalloc) and virtual-memory allocations (hipMemCreate chunks
mapped into one range, scripts/exp/vmm_probe.hip) of the big map, ALTERNATING, every one kept -- so that each draw of either kind
gets physical pages of its own -- with the real kernel timed on each.
    python3 scripts/placement_vmm_draws.py config5 [draws] [chunk MiB, 0 = one chunk]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
n_draws = int(sys.argv[2]) if len(sys.argv) > 2 else 6
chunk_mib = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
probe = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libvmm_probe.so"))
probe.vmm_alloc.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]

wl = bench.WORKLOADS[name]
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
c = 1 if wl["mode"] == "gray" else 3
frames = torch.randint(0, 256, (B,) + wl["hw"] + (c,), device="cuda").float()
which = "end" if wl["mode"] == "gray" else "line_end"


class Raw(object):
    def __init__(self, ptr):
        self.ptr = int(ptr)

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


for _ in range(30):
    pipe.step(frames)
base = getattr(pipe, which)
nbytes = base.numel() * 4
held = [base]
rec = {"workload": name, "map": which, "chunk_MiB": chunk_mib, "plain_ms": [round(kernel_ms(), 4)], "vmm_ms": []}
print("plain 0: %.4f ms" % rec["plain_ms"][0], flush=True)
for i in range(n_draws):
    ptr = C.c_void_p()
    chunk = (chunk_mib << 20) if chunk_mib else (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    rc = probe.vmm_alloc(0, nbytes, chunk, 0, 0, C.byref(ptr))
    if rc:
        sys.exit("vmm_alloc failed: %d" % rc)
    setattr(pipe, which, Raw(ptr.value))
    rec["vmm_ms"].append(round(kernel_ms(), 4))
    print("vmm   %d: %.4f ms" % (i, rec["vmm_ms"][-1]), flush=True)
    new = torch.empty_like(base)
    held.append(new)
    setattr(pipe, which, new)
    rec["plain_ms"].append(round(kernel_ms(), 4))
    print("plain %d: %.4f ms" % (i + 1, rec["plain_ms"][-1]), flush=True)
print(json.dumps(rec), flush=True)
