#!/bin/bash
# rocprofv3 summaries of the bench command (run on the GPU box from the repo root).
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# 60 timed steps: the first ~20 launches after an idle period sit in a power-management transient (profiles/r02/launch_drift.txt);
# a 5-step trace would average only that transient
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-workloads --no-ingest --no-latency --one-stream "$@" > $OUT/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-workloads --no-ingest --no-latency --one-stream "$@" > $OUT/bench_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-workloads --no-ingest --no-latency --one-stream "$@" > $OUT/bench_write.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -20
