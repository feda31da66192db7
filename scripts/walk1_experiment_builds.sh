#!/bin/bash
# ROUND-2 RECORD: the gray strip-walk kernels are no longer in the product tree (scripts/ubench/walk_kernels/README.md);
# run this inside a checkout of the round-2 tree:  git worktree add /tmp/r02 745bae6
# Leave-one-out builds of gray_walk1_kernel (results wrong on purpose), patched scratch copies of csrc/ (never product code).
set -e
cd "$(dirname "$0")/.."
build() {
  local name=$1; local dir=gpurun_exp/src1_$name
  rm -rf $dir; mkdir -p $dir/pysilent_amd/csrc $dir/include
  cp pysilent_amd/csrc/*.h pysilent_amd/csrc/*.hip $dir/pysilent_amd/csrc/; cp include/silent_hip.h $dir/include/
  python3 - "$dir/pysilent_amd/csrc/silent_walk1.h" "$name" <<'PY'
import sys
p, name = sys.argv[1], sys.argv[2]
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
if name == "nopass2":       # unit level only: no window, no completions
    rep("                    if (anchor >= seg_y0 && anchor < seg_y0 + seg_h) {   // wave-uniform", "                    if (anchor >= seg_y0 && anchor < seg_y0 + seg_h && clip_hi == -12345.0f) {")
elif name == "noload":      # the loader issues nothing: consumers alone
    rep("        issue(0, 0);\n        if (n_chunks > 1) issue(1, 1);", "        if (clip_hi == -12345.0f) { issue(0, 0); issue(1, 1); }")
    rep("            if (c + 2 < n_chunks) issue(c + 2, slot2);", "            if (c + 2 < n_chunks && clip_hi == -12345.0f) issue(c + 2, slot2);")
elif name == "nostore":     # no stores of the unit level
    rep("    const bool out_lane = lane >= 4 && lane < 4 + kW1Cols && ox < tab.out_w;", "    const bool out_lane = lane >= 4 && lane < 4 + kW1Cols && ox < tab.out_w && clip_hi == -12345.0f;")
elif name == "noend":       # no end bank (arithmetic + its stores)
    rep("                    if (end_out) {\n                        float acc[K];", "                    if (end_out && clip_hi == -12345.0f) {\n                        float acc[K];")
open(p, "w").write(s)
PY
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off \
      -fno-slp-vectorize -o gpurun_exp/libw1_$name.so $dir/pysilent_amd/csrc/silent_api.hip
}
for v in ${VARIANTS:-nopass2 noload nostore noend}; do build $v & done
wait
ls -la gpurun_exp/libw1_*.so
