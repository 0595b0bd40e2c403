"""Idle time between consecutive kernels of a rocprofv3 kernel trace, grouped by the kernel that ENDS before the gap:
python tools/experiments/gaps.py <dir with p_kernel_trace.csv>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
acc = collections.defaultdict(list)
for a, b in zip(rows[:-1], rows[1:]):
    g = (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3
    if g < 200:                                   # step boundaries (host work) excluded
        acc[a['Kernel_Name'].split('(')[0][-44:]].append(g)
tot = sum(sum(v) for v in acc.values())
print(f'total idle {tot / 1e3:.2f} ms over {sum(len(v) for v in acc.values())} boundaries')
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print(f'{k:46s} n={len(v):4d} avg gap {sum(v) / len(v):6.2f} us  total {sum(v) / 1e3:6.2f} ms')
