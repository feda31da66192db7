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
elif name == "nostore":     # stores are issued but dropped by the range check (num_records = 0)
    rep("orient_out ? lvl_bytes : 0u", "0u")
    rep("line_out ? lvl_bytes : 0u", "0u")
    rep("value_out ? lvl_bytes / 3u : 0u", "0u")
elif name == "noload":      # pixel rows are not fetched inside the loop
    rep("        fetch(mine, row + 2);", "        if (prm.pad == -12345) fetch(mine, row + 2);")
elif name == "norelu":      # the three inner relu / zero-fill selects become plain copies
    rep("g[c] = f2{relu_ok(g[c].x, rok && col0), relu_ok(g[c].y, rok && col1)};", "g[c] = g[c];")
elif name == "storesame":   # every store goes to the first rows of the level (same instructions, L2-resident lines: no DRAM write stream)
    rep("const int ro = (t >= y0 && t < y0 + R && t < H) ? t * W * 12 : kRgb2Out;", "const int ro = (t >= y0 && t < y0 + R && t < H) ? (t & 1) * W * 12 : kRgb2Out;")
    rep("const int ro = rows ? yout * W * 12 : kRgb2Out, rv = rows ? yout * W * 4 : kRgb2Out;", "const int ro = rows ? (yout & 1) * W * 12 : kRgb2Out, rv = rows ? (yout & 1) * W * 4 : kRgb2Out;")
elif name == "loadsame":    # every fetch reads one of the first two rows of the level (L2 hits)
    rep("const int ro = (y >= 0 && y < H) ? y * W * 12 : kRgb2Out;", "const int ro = (y >= 0 && y < H) ? (y & 1) * W * 12 : kRgb2Out;")
elif name == "ntall":       # nt bit on all six stores
    rep("r_orient, st0 + ro, 0, 0);", "r_orient, st0 + ro, 0, 2);")
    rep("r_orient, st1 + ro, 0, 0);", "r_orient, st1 + ro, 0, 2);")
    rep("r_line, st0 + ro, 0, 0);", "r_line, st0 + ro, 0, 2);")
    rep("r_line, st1 + ro, 0, 0);", "r_line, st1 + ro, 0, 2);")
    rep("r_value, sv0 + rv, 0, 0);", "r_value, sv0 + rv, 0, 2);")
    rep("r_value, sv1 + rv, 0, 0);", "r_value, sv1 + rv, 0, 2);")
elif name == "ntol":        # nt bit on the orient / line_end stores
    rep("r_orient, st0 + ro, 0, 0);", "r_orient, st0 + ro, 0, 2);")
    rep("r_orient, st1 + ro, 0, 0);", "r_orient, st1 + ro, 0, 2);")
    rep("r_line, st0 + ro, 0, 0);", "r_line, st0 + ro, 0, 2);")
    rep("r_line, st1 + ro, 0, 0);", "r_line, st1 + ro, 0, 2);")
elif name == "ntld":        # nt bit on the loads too
    rep("r_src, in0 + ro, 0, 0);", "r_src, in0 + ro, 0, 2);")
    rep("r_src, in1 + ro, 0, 0);", "r_src, in1 + ro, 0, 2);")
    rep("r_orient, st0 + ro, 0, 0);", "r_orient, st0 + ro, 0, 2);")
    rep("r_orient, st1 + ro, 0, 0);", "r_orient, st1 + ro, 0, 2);")
    rep("r_line, st0 + ro, 0, 0);", "r_line, st0 + ro, 0, 2);")
    rep("r_line, st1 + ro, 0, 0);", "r_line, st1 + ro, 0, 2);")
elif name == "contig":      # ntol + each x3 store instruction writes one contiguous run (pixels land in the wrong place)
    rep("const int st0 = out0 ? x0 * 12 : kRgb2Out, st1 = out1 ? x1 * 12 : kRgb2Out;", "const int st0 = out0 ? (xw0 + lane - 4) * 12 : kRgb2Out, st1 = out1 ? (xw0 + 56 + lane - 4) * 12 : kRgb2Out;")
    rep("r_orient, st0 + ro, 0, 0);", "r_orient, st0 + ro, 0, 2);")
    rep("r_orient, st1 + ro, 0, 0);", "r_orient, st1 + ro, 0, 2);")
    rep("r_line, st0 + ro, 0, 0);", "r_line, st0 + ro, 0, 2);")
    rep("r_line, st1 + ro, 0, 0);", "r_line, st1 + ro, 0, 2);")
elif name == "contigv":     # contig + the two value stores contiguous as well, nt
    rep("const int st0 = out0 ? x0 * 12 : kRgb2Out, st1 = out1 ? x1 * 12 : kRgb2Out;", "const int st0 = out0 ? (xw0 + lane - 4) * 12 : kRgb2Out, st1 = out1 ? (xw0 + 56 + lane - 4) * 12 : kRgb2Out;")
    rep("const int sv0 = out0 ? x0 * 4 : kRgb2Out, sv1 = out1 ? x1 * 4 : kRgb2Out;", "const int sv0 = out0 ? (xw0 + lane - 4) * 4 : kRgb2Out, sv1 = out1 ? (xw0 + 56 + lane - 4) * 4 : kRgb2Out;")
    rep("r_orient, st0 + ro, 0, 0);", "r_orient, st0 + ro, 0, 2);")
    rep("r_orient, st1 + ro, 0, 0);", "r_orient, st1 + ro, 0, 2);")
    rep("r_line, st0 + ro, 0, 0);", "r_line, st0 + ro, 0, 2);")
    rep("r_line, st1 + ro, 0, 0);", "r_line, st1 + ro, 0, 2);")
    rep("r_value, sv0 + rv, 0, 0);", "r_value, sv0 + rv, 0, 2);")
    rep("r_value, sv1 + rv, 0, 0);", "r_value, sv1 + rv, 0, 2);")
elif name == "swz":         # XCD-aware block order
    rep("const TileCoord tc = locate_tile(tab, blockIdx.x);", "const TileCoord tc = locate_tile(tab, xcd_swizzle(blockIdx.x, gridDim.x));")
elif name == "nopow":       # regulator ratio without log / exp / division
    rep("const f2 rr = {regulator_ratio(bdone.x, prm.rv, prm.root), regulator_ratio(bdone.y, prm.rv, prm.root)};", "const f2 rr = bdone;")
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
