#!/bin/bash
# PMC passes (one counter set per run, kernel trace only) over tools/data_dependence.py <mode>; prints the per-launch
# average of every counter for the tile conv kernel on the RANDOM-data launches (dispatches 5..8 of the script).
MODE=${1:-2}
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1)); rm -rf /tmp/pmc_$i
  echo "== pass $i: $set"
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python /root/repo/tools/data_dependence.py $MODE > /tmp/pmc_$i.log 2>&1 || { echo "pass $i failed or timed out"; tail -3 /tmp/pmc_$i.log; }
  python - "$i" <<'PY'
import csv, glob, sys, collections
f = glob.glob(f'/tmp/pmc_{sys.argv[1]}/**/*counter_collection.csv', recursive=True)
if not f: print('no counter csv for pass', sys.argv[1]); sys.exit()
rows = list(csv.DictReader(open(f[0])))
rows = [r for r in rows if 'conv3x3_tile' in r['Kernel_Name']]
ids = sorted({int(r['Dispatch_Id']) for r in rows})
keep = set(ids[4:8])
acc = collections.defaultdict(list)
for r in rows:
    if int(r['Dispatch_Id']) in keep: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items(): print(f'{k:40s} {sum(v)/len(v):16.0f}')
PY
done <<'SETS'
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_ANY
TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
TA_BUFFER_COALESCED_READ_CYCLES_sum TA_BUFFER_COALESCED_WRITE_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA
SETS
