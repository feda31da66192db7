#!/bin/bash
# Leave-one-out builds of rgb_line_end2_kernel (results wrong on purpose), patched scratch copies of csrc/ (never product code).
#   VARIANTS="noweights nostore ..." scripts/rgb2_experiment_builds.sh ; then scripts/ab_rgb2_libs.sh lib...
set -e
cd "$(dirname "$0")/.."
build() {
  local name=$1; local dir=gpurun_exp/srcr2_$name
  rm -rf $dir; mkdir -p $dir/pysilent_amd/csrc $dir/include
  cp pysilent_amd/csrc/*.h pysilent_amd/csrc/*.hip $dir/pysilent_amd/csrc/; cp include/silent_hip.h $dir/include/
  python3 - "$dir/pysilent_amd/csrc/silent_rgb2.h" "$name" <<'PY'
import sys
p, name = sys.argv[1], sys.argv[2]
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
if name == "noweights":     # the weight stream is never refilled: blocks 0 / 1 requested once, no waits
    rep("            wait_weights();\n            request<(blk + 1) % NB>();", "")
    rep("    ws.template request<0>();", "    ws.template request<0>();\n    ws.template request<1>();")
elif name == "nostore":
    rep("if (orient_out && t >= y0", "if (orient_out && prm.pad == -12345 && t >= y0")
    rep("        if (yout >= y0 && yout < H) {", "        if (yout >= y0 && yout < H && prm.pad == -12345) {")
elif name == "noload":      # pixel rows are not fetched
    rep("        if (row + 1 < NROWS) fetch(nraw, row + 1);", "        if (row + 1 < NROWS && prm.pad == -12345) fetch(nraw, row + 1);")
elif name == "nopow":       # regulator ratio without log / exp / division
    rep("                return prm.rv / pw;", "                return prm.rv * m;")
    rep("                if ((m > 0.0f && m < 7.8886e-31f) || prm.root == 0.0f) pw = powf(m, prm.root);\n                else pw", "                pw")
elif name == "base":
    pass
open(p, "w").write(s)
PY
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off \
      -fno-slp-vectorize -o gpurun_exp/libr2_$name.so $dir/pysilent_amd/csrc/silent_api.hip
  rm -rf $dir
}
for v in ${VARIANTS:-noweights nostore noload nopow}; do build $v & done
wait
ls -la gpurun_exp/libr2_*.so
