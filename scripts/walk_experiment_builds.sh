#!/bin/bash
# ROUND-2 RECORD: the gray strip-walk kernels are no longer in the product tree (scripts/ubench/walk_kernels/README.md);
# run this inside a checkout of the round-2 tree:  git worktree add /tmp/r02 745bae6
# Leave-one-out builds of gray_walk_kernel (results wrong on purpose) made by PATCHING A SCRATCH COPY of csrc/ -- the product
# sources carry no experiment branches.  usage: scripts/walk_experiment_builds.sh ; then on the GPU box
#   scripts/ab_libs.sh pysilent_amd/lib/libsilent_hip.so gpurun_exp/libwalk_noarith.so ...   (AB_ONLY=walk)
set -e
cd "$(dirname "$0")/.."
build() {  # name, python patch body
  local name=$1; local dir=gpurun_exp/src_$name
  rm -rf $dir; mkdir -p $dir/pysilent_amd/csrc $dir/include
  cp pysilent_amd/csrc/*.h pysilent_amd/csrc/*.hip $dir/pysilent_amd/csrc/; cp include/silent_hip.h $dir/include/
  python3 - "$dir/pysilent_amd/csrc/silent_walk.h" "$name" <<'PY'
import re, sys
p, name = sys.argv[1], sys.argv[2]
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
if name == "noarith":      # loads, LDS, barriers and stores as in the product; no stencil arithmetic
    s = re.sub(r"float h0 = wv\[0\] \* La;.*?h1 = __builtin_fmaf\(wv\[4\], Rb, h1\);", "float h0 = a + La * 0.f, h1 = b + Rb * 0.f; (void)Lb; (void)Ra;", s, flags=re.S)
    s = re.sub(r"float v0 = wv\[0\] \* hA\[0\], v1 = wv\[0\] \* hB\[0\];\n#pragma unroll\n\s*for \(int j = 1; j < 5; \+\+j\) \{.*?\}\n", "float v0 = hA[2], v1 = hB[2];\n", s, flags=re.S)
    s = re.sub(r"float a0 = 0.0f, a1 = 0.0f;\n#pragma unroll\n\s*for \(int dy = 0; dy < 3; \+\+dy\)\n#pragma unroll\n\s*for \(int dx = 0; dx < 3; \+\+dx\) \{.*?\}\n", "float a0 = iw[1][1], a1 = iw[1][2];\n", s, flags=re.S)
    s = re.sub(r"for \(int k = 0; k < K; \+\+k\) e0\[k\] = e1\[k\] = 0.0f;.*?e1\[k\] = clip_hi_tf\(relu_tf\(e1\[k\]\), clip_hi\);\n\s*\}", "for (int k = 0; k < K; ++k) { e0[k] = cw[1][1] + k; e1[k] = cw[1][2] + k; }", s, flags=re.S)
elif name == "nostore":    # everything but the stores of the unit level (a never-true runtime condition keeps the arithmetic alive)
    rep("const bool out_lane = lane >= 2 && lane < 62 && colA < tab.out_w;", "const bool out_lane = lane >= 2 && lane < 62 && colA < tab.out_w && clip_hi == -12345.0f;")
elif name == "p2_nostore":     # other levels: everything but the global store of a completed row
    rep("if (lane < gn[g])\n                                pyr[frame_px0", "if (lane < gn[g] && clip_hi == -12345.0f)\n                                pyr[frame_px0")
elif name == "p2_nogather":    # other levels: vertical accumulation only
    rep("if (done < kWalkMaxSlots && anchor >= seg_y0 && anchor < seg_y0 + seg_h) {   // wave-uniform", "if (done < kWalkMaxSlots && anchor >= seg_y0 && anchor < seg_y0 + seg_h && clip_hi == -12345.0f) {")
elif name == "p2_norec":       # other levels: the row record is not read (all-inert): cost of the record reads + branches
    rep("const int meta = __builtin_amdgcn_readfirstlane(cur[g]);", "const int meta = clip_hi == -12345.0f ? __builtin_amdgcn_readfirstlane(cur[g]) : 0;")
elif name.startswith("eu"): # occupancy hint: let the scheduler spend registers on instruction-level parallelism
    n = int(name[2:])
    rep("__global__ __launch_bounds__(walk_threads(G)) void gray_walk_kernel", "__global__ __launch_bounds__(walk_threads(G)) __attribute__((amdgpu_waves_per_eu(%d, %d))) void gray_walk_kernel" % (n, n))
open(p, "w").write(s)
PY
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off \
      -fno-slp-vectorize -o gpurun_exp/libwalk_$name.so $dir/pysilent_amd/csrc/silent_api.hip
}
for v in ${VARIANTS:-noarith nostore eu4 eu3}; do build $v & done
wait
ls -la gpurun_exp/*.so
