#!/bin/bash
# Same-box A/B of two builds of libiodine_hip.so: tools/ab_libs.sh <libA.so> <libB.so> [reps]
# (box-to-box spread of the bench is ~2 %; alternate the two builds on ONE box instead)
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for L in $A $B; do
    IODINE_HIP_LIB=$L python tools/ab_bench.py conv_variant 1 1 train 2 | head -1 | sed "s|^|$(basename $L) |"
    IODINE_HIP_LIB=$L python tools/ab_bench.py conv_variant 1 1 infer 2 | head -1 | sed "s|^|$(basename $L) |"
  done
done
