#!/bin/bash
# Timing-only ablation builds of the 64->64 tile conv (results are WRONG by construction): one library per removed
# ingredient under iodine_amd/ab/, selected with IODINE_HIP_LIB.  tools/run_ablations.sh times them on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p iodine_amd/ab
cp iodine_amd/libiodine_hip.so /tmp/_keep_lib.so
for a in NOINLOAD NOWLOAD NOLDSREAD NOMFMA NOSTORE; do
  touch iodine_amd/csrc/kernels_conv.hip
  IODINE_EXTRA_HIPCC_FLAGS=-DIODINE_ABL_$a python -m iodine_amd.build > /dev/null
  cp iodine_amd/libiodine_hip.so iodine_amd/ab/lib_$a.so
done
touch iodine_amd/csrc/kernels_conv.hip
python -m iodine_amd.build > /dev/null
cp iodine_amd/libiodine_hip.so iodine_amd/ab/lib_BASE.so
