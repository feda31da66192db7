#!/usr/bin/env python3
"""Config 3 (32 x 1080p RGB, 6 levels) for the library selected by SILENT_LIB_PATH: pyramid alone, chain + keypoint tail
alone, whole step; medians of 5 rounds of 10.  Alternate builds with scripts/ab_config3_libs.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd.pipeline import LineEndPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pipe = LineEndPipeline((1080, 1920), mode="rgb", n_levels=6, batch=B, device=0, max_keypoints_per_frame=1 << 16, selection=True,
                       value_map=False, peak_value_map=False)
frames = torch.stack([torch.from_numpy(np.random.default_rng(i).integers(0, 256, (1080, 1920, 3)).astype(np.float32)) for i in range(B)]).cuda()
for _ in range(20):
    pipe.step(frames)
torch.cuda.synchronize()


def timed(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {"pyramid": [], "chain+tail": [], "step": []}
for rnd in range(5):
    res["pyramid"].append(timed(lambda: pipe.run_pyramid(frames)))
    res["chain+tail"].append(timed(pipe.run_filters_keypoints))
    res["step"].append(timed(lambda: pipe.step(frames)))
print("  ".join("%s %.4f" % (k, float(np.median(v))) for k, v in res.items()))
