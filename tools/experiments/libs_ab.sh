#!/bin/bash
# same-box A/B of the library builds in iodine_amd/ab/ (tools/ws_variants.sh build ...) on the whole step:
#   bash tools/experiments/libs_ab.sh [train|infer] [rounds]
MODE=${1:-train}; R=${2:-2}
for i in $(seq 1 $R); do
  for L in iodine_amd/ab/libws_*.so; do
    IODINE_HIP_LIB=$L python tools/ab_bench.py zigzag 1 1 $MODE 3 | head -1 | sed "s|^|$(basename $L .so) |"
  done
done
