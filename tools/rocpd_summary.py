#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / avg / min / max / share.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        'select name, count(*), sum(duration), avg(duration), min(duration), max(duration), '
        'max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) '
        'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f'# rocprofv3 --kernel-trace --stats summary of `{path}`\n')
    print('| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | grid | wg |')
    print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    for n, k, tot, avg, mn, mx, vg, ag, sg, lds, gx, wx in rows:
        short = n if len(n) < 110 else n[:107] + '...'
        print(f'| `{short}` | {k} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | '
              f'{100.0 * tot / total:.2f} | {vg} | {ag} | {sg} | {lds} | {gx} | {wx} |')
    print(f'\ntotal kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches')


if __name__ == '__main__':
    main(sys.argv[1])
