#!/bin/bash
# RGB chain kernel time (config 3) for several library builds, alternating: scripts/ab_rgb2_libs.sh "<knobs>" lib1 lib2 ...
KNOBS=$1; shift
for i in 1 2; do
for L in "$@"; do
  printf "%-24s " $(basename $L); SILENT_LIB_PATH=$PWD/$L timeout -k 10 200 python scripts/ab_rgb_chain.py 32 $KNOBS 2>&1 | grep TUNE_RGB | sed 's/TUNE_RGB //; s/ of 40 B.*//' | tr '\n' '|'; echo
done; done
