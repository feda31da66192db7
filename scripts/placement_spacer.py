import sys; sys.path.insert(0, "/root/repo")
import torch, bench
for name in ("config2", "config5", "config3"):
    wl = bench.WORKLOADS[name]; B = wl["frames"]; c = 1 if wl["mode"] == "gray" else 3
    for sp in (0.0, 12.0, 24.0):
        pipe = bench.make_pipeline(wl, B, 0, None)
        frames = torch.randint(0, 256, (B,) + wl["hw"] + (c,), device="cuda").float()
        print(name, "spacer", sp, pipe.tune_placement(frames, tries=8, spacer_gib=sp, budget_s=8), flush=True)
        del pipe, frames
        torch.cuda.empty_cache()
