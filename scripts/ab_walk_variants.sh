#!/bin/bash
# ROUND-2 RECORD: the gray strip-walk kernels are no longer in the product tree (scripts/ubench/walk_kernels/README.md);
# run this inside a checkout of the round-2 tree:  git worktree add /tmp/r02 745bae6
# alternating A/B of library builds: usage scripts/ab_walk_variants.sh <variant name of ab_pass.py> <gray opts> lib1 lib2 ...
VAR=$1; OPTS=$2; shift; shift
export AB_ONLY="$VAR"
for i in 1 2 3; do
for L in "$@"; do
  printf "%-28s " $(basename $L); AB_BASE_OPTS=$OPTS SILENT_AB_WALK_OPTS=$OPTS SILENT_LIB_PATH=$PWD/$L timeout -k 10 200 python scripts/ab_pass.py 2>&1 | grep -E "^$VAR |kernel alone" | sed 's/GB\/s.*//; s/| 1 GiB.*//' | tr '\n' ' '; echo
done; done
