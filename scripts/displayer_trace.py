import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pysilent_amd.recognition_testing import LineEndDisplayer
h, w = (1080, 1920) if len(sys.argv) > 1 and sys.argv[1] == "1080" else (480, 640)
d = LineEndDisplayer()
f = np.random.default_rng(0).integers(0, 256, (h, w, 3)).astype(np.uint8)
for _ in range(12):
    d.callback(f)
