#!/bin/bash
# GPU-box side: bash tools/wino_ab.sh [rounds] [modes] -> per-kernel average durations of the interleaved A/B (tools/wino_ab.py)
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/winoab
rocprofv3 --kernel-trace --stats -d /tmp/winoab -o t --output-format csv -- python $REPO/tools/wino_ab.py "$@" > /tmp/winoab.log 2>&1 || tail -5 /tmp/winoab.log
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/winoab/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if 'conv3x3' in k:
        v2 = v[2:] if len(v) > 4 else v
        print(f'{sum(v2) / len(v2):9.1f} us avg  (min {min(v2):.1f}, n {len(v)})  {k[:100]}')
PY
