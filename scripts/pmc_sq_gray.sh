#!/bin/bash
# SQ counters of gray_stream_kernel alone (config 2 through bench.py), three passes of <= 8 counters.
#   usage (GPU box, repo root): scripts/pmc_sq_gray.sh <tag> [bench args]
set -o pipefail
TAG=${1:-gray}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P1="SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY"
P2="SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
P3="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-include-regex gray_stream --output-format csv -d $OUT/p$i -o p$i -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-latency --no-ingest --no-side-workloads "$@" > $OUT/p$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for c, v in sorted(acc.items()): print("%s %-32s %.4g per launch (%d launches)" % (p, c, v / max(n[c], 1), n[c]))
PY
