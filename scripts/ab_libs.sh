#!/bin/bash
# A/B builds (or knob settings) of the library on the same box, alternating.
# usage: scripts/ab_libs.sh <lib.so>[@SILENT_GRAY_OPTS] <lib.so>[@opts] ... ; ROUNDS=3
N=${ROUNDS:-3}
for i in $(seq $N); do
  for E in "$@"; do
    L=${E%@*}; O=0; [[ "$E" == *@* ]] && O=${E#*@}
    printf "%-34s " "$(basename $L)@$O"
    AB_BASE_OPTS=$O SILENT_LIB_PATH=$PWD/$L python scripts/ab_pass.py 2>&1 | grep -E "kernel alone|^stream " | sed 's/stream kernel alone: /kernel /; s/|.*//; s/min.*//' | tr '\n' ' '; echo
  done
done
