#!/bin/bash
# SQ counter passes over the bench command (run on the GPU box from the repo root): where do the waves of the
# dominant kernel spend their cycles?   usage: scripts/pmc_sq.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-sq}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $OUT/p1 -o p1 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-ingest "$@" > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -o p2 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-ingest "$@" > $OUT/p2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_SALU"): n[k] += 1
        for k, d in acc.items():
            if "silent" not in k: continue
            print(p, k, "launches", n[k])
            for c, v in sorted(d.items()): print("   %-24s %.4g per launch" % (c, v / max(n[k], 1)))
PY
