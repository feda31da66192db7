#!/bin/bash
# alternating A/B of library builds on the walk path: usage scripts/ab_walk_variants.sh <gray opts> lib1 lib2 ...
OPTS=$1; shift
export AB_ONLY="walk"
for i in 1 2; do
for L in "$@"; do
  printf "%-32s " $(basename $L); AB_BASE_OPTS=$OPTS SILENT_AB_WALK_OPTS=$OPTS SILENT_LIB_PATH=$PWD/$L timeout -k 10 200 python scripts/ab_pass.py 2>&1 | grep -E "^walk |kernel alone" | sed 's/GB\/s.*//; s/| 1 GiB.*//' | tr '\n' ' '; echo
done; done
