#!/bin/bash
# Leave-one-out builds of pyramid_walk3_kernel (results wrong on purpose), patched scratch copies of csrc/ (never product code).
set -e
cd "$(dirname "$0")/.."
build() {
  local name=$1; local dir=gpurun_exp/src3_$name
  rm -rf $dir; mkdir -p $dir/pysilent_amd/csrc $dir/include
  cp pysilent_amd/csrc/*.h pysilent_amd/csrc/*.hip $dir/pysilent_amd/csrc/; cp include/silent_hip.h $dir/include/
  python3 - "$dir/pysilent_amd/csrc/silent_walk_rgb.h" "$name" <<'PY'
import sys
p, name = sys.argv[1], sys.argv[2]
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
if name == "noconsume":     # consumers only meet the barriers: the loader / barrier protocol alone
    rep("        if (live) {\n#pragma unroll\n            for (int r = 0; r < kWalkCH; ++r) {", "        if (live && tab.wx[0] == -12345.0f) {\n#pragma unroll\n            for (int r = 0; r < kWalkCH; ++r) {")
elif name == "noload":      # the loader issues nothing (and waits for nothing): the consumers alone, on whatever the ring holds
    rep("        issue(0, 0);\n        if (n_chunks > 1) issue(1, 1);", "        if (tab.wx[0] == -12345.0f) { issue(0, 0); issue(1, 1); }")
    rep("            if (c + 2 < n_chunks) issue(c + 2, slot2);", "            if (c + 2 < n_chunks && tab.wx[0] == -12345.0f) issue(c + 2, slot2);")
elif name == "nopass2":     # unit level only
    rep("                    if (!(meta & 128)) return;                  // wave-uniform: this source row carries no tap of level g", "                    if (!(meta & 128) || tab.wx[0] != -12345.0f) return;")
elif name == "nostore0":    # no stores of the unit level
    rep("                    if (out_lane) {\n                        typedef float nf2", "                    if (out_lane && tab.wx[0] == -12345.0f) {\n                        typedef float nf2")
open(p, "w").write(s)
PY
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off \
      -fno-slp-vectorize -o gpurun_exp/libw3_$name.so $dir/pysilent_amd/csrc/silent_api.hip
}
for v in ${VARIANTS:-noconsume noload nopass2 nostore0}; do build $v & done
wait
ls -la gpurun_exp/libw3_*.so
