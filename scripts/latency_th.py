import sys, os, time; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pysilent_amd.recognition_testing import LineEndDisplayer
d = LineEndDisplayer()
f = np.random.default_rng(0).integers(0, 256, (480, 640, 3)).astype(np.uint8)
for _ in range(20):
    d.callback(f, copy=False)
fd = d._native[1]
np.copyto(fd.frame_buffer, f)
ts, busy = [], []
for _ in range(300):
    t0 = time.perf_counter(); fd.step(fd.frame_buffer); ts.append((time.perf_counter() - t0) * 1e3)
tt = []
for _ in range(100):
    t0 = time.perf_counter(); fd.step(fd.frame_buffer, timing=True); tt.append((time.perf_counter() - t0) * 1e3); busy.append(fd.gpu_ms)
print("SILENT_RGB_OPTS=%s  in-place p50 %.4f ms (p99 %.4f)   with events + stream synchronisation %.4f   gpu busy %.4f" % (
    os.environ.get("SILENT_RGB_OPTS", "(unset)"), np.percentile(ts, 50), np.percentile(ts, 99), np.percentile(tt, 50), np.median(busy)))
