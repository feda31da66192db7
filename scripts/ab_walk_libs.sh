#!/bin/bash
# pyramid walk + step for several library builds, alternating: WL=config3 ROUNDS=2 scripts/ab_walk_libs.sh lib1.so lib2.so ...
for i in $(seq ${ROUNDS:-2}); do
for L in "$@"; do
  printf "%-28s %s " $(basename $L) ${WL:-config3}; SILENT_LIB_PATH=$PWD/$L timeout -k 10 200 python scripts/ab_walk.py ${WL:-config3} 2>&1 | tail -1
done; done
