#!/bin/bash
# Compile-time variants of the Winograd EXPERIMENT (tools/experiments/kernels_wino.hip) as separate libraries (container):  tools/wino_variants.sh name:-DFLAG ...
# -> iodine_amd/ab/libwino_<name>.so; run one with IODINE_HIP_LIB=... (tools/wino_ab.sh)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
AB=$REPO/iodine_amd/ab
mkdir -p $AB
B=$REPO/iodine_amd/csrc/build
SRC=${WINO_SRC:-$REPO/tools/experiments/kernels_wino.hip}       # WINO_SRC=tools/experiments/kernels_wino2.hip: the round-4 variant
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -x hip -c $SRC -o /tmp/wino_$name.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DIODINE_WITH_WINO -x hip -c $REPO/iodine_amd/csrc/iodine_api.cpp -o /tmp/wino_api.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $AB/libwino_$name.so /tmp/wino_$name.o /tmp/wino_api.o $B/kernels_conv.hip.o $B/kernels_convws.hip.o $B/kernels_out.hip.o $B/kernels_pixel.hip.o $B/kernels_misc.hip.o $B/kernels_train.hip.o $B/kernels_refine.hip.o $B/kernels_refbwd.hip.o $B/kernels_generic.hip.o || exit 1
  echo built $name "($flags)"
done
