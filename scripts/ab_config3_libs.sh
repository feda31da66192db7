#!/bin/bash
# config 3 parts for several library builds, alternating: scripts/ab_config3_libs.sh lib1.so lib2.so ...   (ROUNDS=2)
for i in $(seq ${ROUNDS:-2}); do
for L in "$@"; do
  printf "%-28s " $(basename $L); SILENT_LIB_PATH=$PWD/$L timeout -k 10 200 python scripts/ab_config3_parts.py 2>&1 | tail -1
done; done
