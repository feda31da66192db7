import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
from pysilent_amd import _lib, _runtime
wl = bench.WORKLOADS["reference_layout_gray"]; B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
frames = torch.randint(0, 256, (B,) + wl["hw"] + (1,), device="cuda").float()
def timed(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
ctx = _runtime.get_context(0)
for knob in (0, 1, 0, 1):
    ctx.set_tuning(_lib.TUNE_PYRAMID, knob)
    for _ in range(20): pipe.step(frames)
    print("pyramid knob %d (%s): step %.4f ms   pyramid alone %.4f ms" % (knob, "unit + region kernels" if knob else "stream plans", np.median([timed(lambda: pipe.step(frames)) for _ in range(5)]), np.median([timed(lambda: pipe.run_pyramid(frames)) for _ in range(5)])))
ctx.set_tuning(_lib.TUNE_PYRAMID, 0)
a = pipe.outputs(); pipe.step(frames); torch.cuda.synchronize()
x = {k: a[k].data.clone() for k in ("pyramid", "cs", "end")}
ctx.set_tuning(_lib.TUNE_PYRAMID, 1); pipe.step(frames); torch.cuda.synchronize()
print("bit-identical to unit + region:", all(torch.equal(x[k].view(torch.int32), pipe.outputs()[k].data.view(torch.int32)) for k in x))
