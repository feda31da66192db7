#!/usr/bin/env python3
"""What does this box's HBM give a plain streaming kernel?  write-only (fill), read-only (sum), copy, and a
1 : 3.2 read : write mix like the stream kernel's (read 1 float, write 4 floats... per element)."""
import numpy as np, torch
n = 1 << 28
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
def t(fn, reps=8):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:]))
gb = n * 4 / 1e9
print("fill  (write only)   %.0f GB/s" % (gb / t(lambda: a.fill_(1.0)) * 1e3))
print("sum   (read only)    %.0f GB/s" % (gb / t(lambda: a.sum()) * 1e3))
print("copy  (1 r : 1 w)    %.0f GB/s" % (2 * gb / t(lambda: b.copy_(a)) * 1e3))
q = a[: n // 4]
w4 = b.view(n // 4, 4)
print("expand (1 r : 4 w)   %.0f GB/s" % ((gb / 4 + gb) / t(lambda: w4.copy_(q[:, None].expand(n // 4, 4))) * 1e3))
