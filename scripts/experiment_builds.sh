#!/bin/bash
# Leave-one-out builds of the ROUND-1 gray_stream_kernel (results are wrong on purpose): which resource bounds it?
#   1 = pass 1 stores nothing, 2 = pass 1 does no arithmetic, 3 = 2 without the frame loads, ... (see the header copy).
# The experiment branches live in scripts/ubench/silent_conv_experiments_r01.h, not in the product sources: this script
# compiles a scratch copy of csrc/ with that header in place of silent_conv.h.
# usage (in the container): scripts/experiment_builds.sh ; then on the GPU box: SILENT_LIB_PATH=... python scripts/ab_pass.py
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_exp/src/pysilent_amd/csrc gpurun_exp/src/include
cp pysilent_amd/csrc/*.h pysilent_amd/csrc/*.hip gpurun_exp/src/pysilent_amd/csrc/
cp include/silent_hip.h gpurun_exp/src/include/
cp scripts/ubench/silent_conv_experiments_r01.h gpurun_exp/src/pysilent_amd/csrc/silent_conv.h
for e in ${EXPERIMENTS:-1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off \
      -fno-slp-vectorize -DSILENT_EXPERIMENT=$e -o gpurun_exp/libsilent_exp$e.so gpurun_exp/src/pysilent_amd/csrc/silent_api.hip
done
ls -la gpurun_exp
