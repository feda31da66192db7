import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "reference_layout"]; B = wl["frames"]
pipe = bench.make_pipeline(wl, B, 0, None)
frames = torch.randint(0, 256, (B,) + wl["hw"] + (3,), device="cuda").float()
for _ in range(6): pipe.run_pyramid(frames)
torch.cuda.synchronize()
