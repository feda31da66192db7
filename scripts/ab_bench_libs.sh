#!/bin/bash
# Alternating A/B of library builds through bench.py itself (contract timing: settle loop, K steps, steady state, kernel events).
# usage: WORKLOAD=config5 ROUNDS=2 scripts/ab_bench_libs.sh gpurun_exp/lib_a.so gpurun_exp/lib_b.so ...
N=${ROUNDS:-2}; WL=${WORKLOAD:-config2}
for i in $(seq $N); do
  for L in "$@"; do
    printf "%-28s %s  " "$(basename $L)" "$WL"
    SILENT_LIB_PATH=$PWD/$L python bench.py --workload $WL --steps 30 --warmup 5 --no-side-workloads --no-cpu-baseline --no-ingest --no-latency 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step %.4f  steady %.4f  kernel %.4f ms  frac %.3f  settle %d' % (d['ms_per_step'], d['steady_state']['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['settle_steps_run']))"
  done
done
