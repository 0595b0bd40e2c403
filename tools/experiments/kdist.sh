# duration distribution of kernels matching a pattern in a short bench run: bash tools/experiments/kdist.sh <pattern> [bench args]
PAT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/kd; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kd -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-configs --no-sustain "$@" > /dev/null 2>&1
cd $R
python - "$PAT" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/kd/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:70], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', ''))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items()):
    v.sort()
    # cluster by duration (layers differ by ~4x)
    print(k, 'n', len(v), 'min %.1f p25 %.1f median %.1f p75 %.1f max %.1f' % (v[0], v[len(v)//4], v[len(v)//2], v[3*len(v)//4], v[-1]))
PY
