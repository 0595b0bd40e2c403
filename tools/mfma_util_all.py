#!/usr/bin/env python3
"""Matrix-pipe utilisation and shader clock of EVERY kernel of a rocprofv3 pass that collected SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE with
--kernel-trace (tools/pmc_traffic.sh writes such a pass to gpurun_out/pmc_<tag>/SQ_VALU_MFMA_BUSY_CYCLES):
    python tools/mfma_util_all.py gpurun_out/pmc_r05/SQ_VALU_MFMA_BUSY_CYCLES
util = busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / kernel duration."""
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: dict(busy=[], gui=[], clk=[], us=[]))
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    kt = f.replace('counter_collection', 'kernel_trace')
    dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(kt))} if os.path.exists(kt) else {}
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0][-70:]
        a = acc[n]
        if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES': a['busy'].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            a['gui'].append(float(r['Counter_Value']))
            d = dur.get(r['Dispatch_Id'], 0)
            if d > 0: a['clk'].append(float(r['Counter_Value']) / 8.0 / d); a['us'].append(d / 1e3)
print('| kernel | launches | avg us | matrix pipe busy | clock GHz | rate of the 2.4 GHz peak |\n|---|---|---|---|---|---|')
rows = []
for n, a in acc.items():
    if not a['busy'] or not a['gui'] or not a['us']: continue
    busy, gui = sum(a['busy']) / len(a['busy']), sum(a['gui']) / len(a['gui'])
    util, clk, us = busy / (gui / 8.0 * 1024.0), sum(a['clk']) / len(a['clk']), sum(a['us']) / len(a['us'])
    rows.append((us * len(a['us']), n, len(a['us']), us, util, clk))
for _, n, k, us, util, clk in sorted(rows, reverse=True)[:25]:
    print(f'| `{n}` | {k} | {us:.1f} | {util:.3f} | {clk:.2f} | {util * clk / 2.4:.3f} |')
