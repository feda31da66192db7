#!/bin/bash
# Builds of the library with one SILENT_EXPERIMENT switch each (results are wrong on purpose): which resource
# bounds gray_stream_kernel?   1 = pass 1 stores nothing, 2 = pass 1 does no arithmetic, 3 = 2 without the frame loads.
# usage (in the container): scripts/experiment_builds.sh ; then on the GPU box: SILENT_LIB_PATH=... python scripts/ab_pass.py
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_exp
for e in ${EXPERIMENTS:-1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off \
      -fno-slp-vectorize -DSILENT_EXPERIMENT=$e -o gpurun_exp/libsilent_exp$e.so pysilent_amd/csrc/silent_api.hip
done
ls -la gpurun_exp
