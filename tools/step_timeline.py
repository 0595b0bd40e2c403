#!/usr/bin/env python3
"""Timeline of ONE steady-state step out of a rocprofv3 kernel trace (rocpd SQLite): where the time between the kernels goes.

    python tools/step_timeline.py x_results.db [anchor-kernel-substring] [--list]

A step is delimited by consecutive launches of the anchor kernel (default adam_multi_kernel: once per training step).  Prints the
step's wall time, the sum of kernel durations, the idle time between kernels bucketed by the size of the gap, and per kernel name
the launches, their time and the idle time in FRONT of them (gap to the end of the previous kernel); --list dumps every launch."""
import collections
import sqlite3
import sys


def main(path, anchor='adam_multi_kernel', dump=False):
    c = sqlite3.connect(path)
    rows = c.execute('select name, start, end from kernels order by start').fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(idx) < 3:
        sys.exit(f'anchor {anchor!r}: {len(idx)} launches')
    a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]                # a step in the middle of the run
    step = rows[a:b]
    wall = (rows[b][1] - rows[a][1]) / 1e3
    busy = sum(e - s for _, s, e in step) / 1e3
    print(f'step of {len(step)} launches: wall {wall:.1f} us, kernels {busy:.1f} us, idle {wall - busy:.1f} us')
    per = collections.OrderedDict()
    buckets = collections.Counter()
    prev_end = None
    for n, s, e in step + [rows[b]]:
        gap = 0.0 if prev_end is None else max(0.0, (s - prev_end) / 1e3)
        prev_end = max(prev_end or e, e)
        if (n, s, e) == rows[b]:
            buckets['<2' if gap < 2 else '2-5' if gap < 5 else '5-20' if gap < 20 else '>=20'] += gap
            break
        k = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-64:]
        d = per.setdefault(k, [0, 0.0, 0.0])
        d[0] += 1; d[1] += (e - s) / 1e3; d[2] += gap
        buckets['<2' if gap < 2 else '2-5' if gap < 5 else '5-20' if gap < 20 else '>=20'] += gap
        if dump:
            print(f'{(s - rows[a][1]) / 1e3:10.1f} us  +{gap:6.1f} idle  {(e - s) / 1e3:8.1f} us  {k}')
    print('idle by gap size (us):', {k: round(v, 1) for k, v in buckets.items()})
    print('| kernel | launches | kernel us | idle in front us |')
    print('|---|---|---|---|')
    for k, (n, t, g) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f'| `{k}` | {n} | {t:.1f} | {g:.1f} |')


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if a != '--list']
    main(args[0], *(args[1:2]), dump='--list' in sys.argv)
