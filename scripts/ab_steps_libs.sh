#!/bin/bash
# whole-step A/B of library builds, alternating: scripts/ab_steps_libs.sh "config2 config3" lib1 lib2 ...
WLS=$1; shift
for i in 1 2 3; do
for L in "$@"; do
  printf "%-20s " $(basename $L); for w in $WLS; do SILENT_LIB_PATH=$PWD/$L timeout -k 10 200 python scripts/ab_steps.py $w 2>&1 | grep "step:" | tr '\n' '|'; done; echo
done; done
