#!/bin/bash
# SQ / LDS counter passes over tools/conv_time.py <mode> (one counter set per rocprofv3 run, kernel trace only)
REPO=$(cd "$(dirname "$0")/.." && pwd)
MODE=${1:-10}
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1)); rm -rf /tmp/wspmc_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/wspmc_$i -o p -- python $REPO/tools/conv_time.py $MODE > /tmp/wspmc_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/wspmc_$i.log; }
  python - "$i" <<'PY'
import csv, glob, sys, collections
f = glob.glob(f'/tmp/wspmc_{sys.argv[1]}/**/*counter_collection.csv', recursive=True)
if not f: print('no counter csv for pass', sys.argv[1]); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n = r['Kernel_Name']
    if 'conv3x3_ws' in n or 'conv3x3_tile_f16x3' in n:
        acc[n.split('(')[0][-30:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, ' '.join(f'{c}={sum(v[2:])/max(1,len(v[2:])):.4g}' for c, v in d.items()))
PY
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
SETS
