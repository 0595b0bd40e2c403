#!/usr/bin/env python3
"""Where the wave cycles of each kernel go: python tools/sq_split.py <dir of a rocprofv3 --pmc pass (csv)> [min share of total cycles]

The pass collects SQ_WAVE_CYCLES, SQ_WAIT_ANY (wave parked: s_waitcnt / s_barrier), SQ_WAIT_INST_ANY (issue stall: dependency / pipe busy),
SQ_ACTIVE_INST_ANY (an instruction issuing) - the three are disjoint and add up to about SQ_WAVE_CYCLES (MI355X_MICROARCH.md, "rocprofv3 PMC
slots") - plus SQ_WAIT_INST_LDS, SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_LDS and SQ_LDS_BANK_CONFLICT.  Prints, per kernel, launches and each
counter as a share of the kernel's wave cycles (mean over launches)."""
import collections
import csv
import glob
import os
import sys


def main(d, min_share=0.01):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    tot = sum(sum(v.get('SQ_WAVE_CYCLES', [0])) for v in acc.values()) or 1.0
    cols = ['SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_LDS_BANK_CONFLICT']
    print('| kernel | launches | share of all wave cycles | ' + ' | '.join(c.replace('SQ_', '') for c in cols) + ' |')
    print('|---|---|---|' + '---|' * len(cols))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0]))):
        wc = v.get('SQ_WAVE_CYCLES')
        if not wc or sum(wc) / tot < min_share:
            continue
        m = sum(wc) / len(wc)
        cells = []
        for c in cols:
            x = v.get(c)
            cells.append(f'{(sum(x) / len(x)) / m:.3f}' if x else '-')
        print(f'| `{k}` | {len(wc)} | {sum(wc) / tot:.3f} | ' + ' | '.join(cells) + ' |')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.01)
