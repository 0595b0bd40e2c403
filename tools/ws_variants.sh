#!/bin/bash
# Same-box A/B of compile-time variants of kernels_convws.hip.  Build side (container):  tools/ws_variants.sh build name:-DFLAG ...
# Run side (GPU box):  tools/ws_variants.sh run [modes...]  -> per variant the steady-state kernel times under rocprofv3
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
AB=$REPO/iodine_amd/ab
if [ "$1" = build ]; then
  shift; mkdir -p $AB; rm -f $AB/libws_*.so
  B=$REPO/iodine_amd/csrc/build
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -x hip -c $REPO/iodine_amd/csrc/kernels_convws.hip -o /tmp/ws_$name.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $AB/libws_$name.so /tmp/ws_$name.o $B/kernels_conv.hip.o $B/kernels_out.hip.o $B/kernels_pixel.hip.o $B/kernels_misc.hip.o $B/kernels_train.hip.o $B/kernels_refine.hip.o $B/iodine_api.cpp.o || exit 1
    echo built $name "($flags)"
  done
  exit 0
fi
shift
cd /tmp && export TMPDIR=/tmp
for lib in $AB/libws_*.so; do
  name=$(basename $lib .so); rm -rf /tmp/wsv_$name
  IODINE_HIP_LIB=$lib rocprofv3 --kernel-trace -d /tmp/wsv_$name -o t --output-format csv -- python $REPO/tools/conv_time.py "$@" > /tmp/wsv_$name.log 2>&1
  python - $name <<'PY'
import csv, glob, sys, collections
f = glob.glob(f'/tmp/wsv_{sys.argv[1]}/**/*kernel_trace.csv', recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n = r['Kernel_Name']
    if 'conv3x3_' in n: acc[n[:44]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(sys.argv[1], ' | '.join(f'{k.split("(")[0][-28:]}: ' + ' '.join(f'{sum(v[i:i+6][2:])/4:.0f}' for i in range(0, len(v), 6)) for k, v in acc.items()))
PY
done
