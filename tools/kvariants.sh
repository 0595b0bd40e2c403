#!/bin/bash
# Compile-time variants of ONE kernel file as side libraries (container):  tools/kvariants.sh kernels_refl0.hip name:-DFLAG name2:"-DA -DB" ...
# -> iodine_amd/ab/libvar_<name>.so (the other objects from iodine_amd/csrc/build); run with IODINE_HIP_LIB=... (tools/ab_libs.py, tools/cat_times.py)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
AB=$REPO/iodine_amd/ab; mkdir -p $AB
B=$REPO/iodine_amd/csrc/build
F=$1; shift
OTHERS=$(ls $B/*.o | grep -v "/$F.o")
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -x hip -c $REPO/iodine_amd/csrc/$F -o /tmp/var_$name.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $AB/libvar_$name.so /tmp/var_$name.o $OTHERS || exit 1
  echo built $name "($flags)"
done
