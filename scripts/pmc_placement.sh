#!/bin/bash
# Memory-side counters on a fast and a slow allocation of the big write-streamed map.
#   usage (GPU box, repo root): scripts/pmc_placement.sh <workload> <kernel substring> [first pass] [last pass]
# One rocprofv3 process per pass (<= 4 TCC counters each); each finds its own fast / slow pair (scripts/placement_pmc.py).
set -o pipefail
WL=${1:-config5}; KSUB=${2:-gray_stream}; FIRST=${3:-1}; LAST=${4:-8}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_placement_$WL
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PASS[1]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
PASS[2]="TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum"
PASS[3]="TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum"
PASS[4]="TCC_EA0_WRREQ_max TCC_EA0_WRREQ_min TCC_EA0_WRREQ_STALL_max6 TCC_EA0_WRREQ_STALL_min6"
PASS[5]="TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
PASS[6]="TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
# (a pass of GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_TC_BUSY never returned on the pool's boxes -- killed after 7 silent minutes,
# round 6 -- GRBM counters are left out)
PASS[7]="TCC_REQ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum"
PASS[8]="TCC_BUSY_sum TCC_CYCLE_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum"
for i in $(seq $FIRST $LAST); do
  P=${PASS[$i]}
  rocprofv3 -E $REPO/scripts/pmc/placement_extra_counters.yaml --pmc $P --kernel-include-regex $KSUB --output-format csv -d $OUT/p$i -o p$i -- \
    python3 $REPO/scripts/placement_pmc.py $WL $OUT/p$i.json --tries 8 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; continue; }
  python3 $REPO/scripts/placement_pmc_join.py $OUT/p$i $OUT/p$i.json $KSUB | tee -a $OUT/summary.txt
done
