#!/usr/bin/env python3
"""Round 6 placement experiment 6: one process, N draws of the big map (all kept), and on EVERY draw the real kernel beside the
synthetic writers (scripts/exp/vmm_probe.hip) -- the K map alone with the kernel's tile geometry, a linear fill, the three write
streams together, the kernel's streams with the frame reads -- so that the columns can be compared draw by draw.
    python3 scripts/placement_table.py config5 [draws]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
n_draws = int(sys.argv[2]) if len(sys.argv) > 2 else 12
probe = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libvmm_probe.so"))
probe.wp2_run.restype = C.c_float
probe.wp2_run.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_longlong, C.c_int, C.c_int]
probe.wp_run.restype = C.c_float
probe.wp_run.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]

wl = bench.WORKLOADS[name]
assert wl["mode"] == "gray"
B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
frames = torch.randint(0, 256, (B,) + wl["hw"] + (1,), device="cuda").float()
h, w = wl["hw"]
K = wl["n_orient"]


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


def synth(buf, flags):
    return float(probe.wp2_run(C.c_void_p(buf.data_ptr()), C.c_void_p(pipe.cs.data_ptr()), C.c_void_p(pipe._pyrs[0].data_ptr()),
                               C.c_void_p(frames.data_ptr()), K, w, h, B, pipe.frame_px, flags, 6))


def linear(buf):
    row_bytes = w * 4 * K
    return float(probe.wp_run(C.c_void_p(buf.data_ptr()), row_bytes, h, B, row_bytes // 4, 1, 1, 6))


def copy_ms(buf):
    flat = buf.view(-1)
    n = flat.numel() // 2
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    flat[n:2 * n].copy_(flat[:n])
    a.record()
    for _ in range(3):
        flat[n:2 * n].copy_(flat[:n])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 3


for _ in range(30):
    pipe.step(frames)
held = [pipe.end]
rows = []
print("draw   real kernel   K map alone   linear fill   3 write streams   + frame reads   copy half->half (ms)", flush=True)
for i in range(n_draws):
    if i:
        held.append(torch.empty_like(held[0]))
        pipe.end = held[-1]
    buf = held[-1]
    r = [kernel_ms(), synth(buf, 0), linear(buf), synth(buf, 6), synth(buf, 7), copy_ms(buf), kernel_ms(4, 8),
         synth(buf, 6 | 32), synth(buf, 6 | 32 | 64), synth(buf, 64), synth(buf, 7 | 32), synth(buf, 7 | 32 | 64), synth(buf, 2 | 32), synth(buf, 4 | 32)]
    rows.append([round(v, 4) for v in r])
    print("%4d   %.4f        %.4f        %.4f        %.4f            %.4f          %.4f    (real again %.4f)" % (i, *r[:7]), flush=True)
    print("       block-wide runs (a wave writes a whole 224-px tile row): 3 streams, 1-ch maps so %.4f   + K map so %.4f   K map alone so %.4f   "
          "with reads: 1-ch so %.4f  all so %.4f   K + CS(block) %.4f   K + pyr(block) %.4f" % tuple(r[7:]), flush=True)
print(json.dumps({"workload": name, "columns": ["real", "k_map_alone", "linear_fill", "three_write_streams", "with_frame_reads", "copy", "real_again",
                                                "three_streams_1ch_blockwide", "three_streams_all_blockwide", "k_alone_blockwide",
                                                "with_reads_1ch_blockwide", "with_reads_all_blockwide", "k_cs_blockwide", "k_pyr_blockwide"],
                  "rows": rows}), flush=True)
