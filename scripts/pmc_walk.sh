#!/bin/bash
# SQ counters of the pyramid walk alone (scripts/walk_only.py <workload>): usage (GPU box, repo root): scripts/pmc_walk.sh <workload>
set -o pipefail
WL=${1:-reference_layout}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_walk_$WL; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/p1 -o p1 -- python3 $REPO/scripts/walk_only.py $WL > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/p2 -o p2 -- python3 $REPO/scripts/walk_only.py $WL > $OUT/p2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_GDS SQ_WAVES_EQ_64 --output-format csv -d $OUT/p3 -o p3 -- python3 $REPO/scripts/walk_only.py $WL > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
        for k, d in acc.items():
            if "walk3" not in k and "border" not in k: continue
            print(p, k)
            for c, v in sorted(d.items()): print("   %-24s %.4g per launch (%d launches)" % (c, v / max(n[k][c], 1), n[k][c]))
PY
