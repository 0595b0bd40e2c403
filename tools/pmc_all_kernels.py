#!/usr/bin/env python3
"""HBM traffic per launch of EVERY kernel of the bench step from the FETCH_SIZE / WRITE_SIZE passes tools/pmc_traffic.sh
already ran (one counter per rocprofv3 run, kernel trace in the same run): bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024
[gfx950: 16 B/lane reads are tallied at half, MI355X_MICROARCH.md "HBM"], duration from the kernel trace of the FETCH pass.

    python tools/pmc_all_kernels.py gpurun_out/pmc_r02 > profiles/r02_kernel_traffic.md
"""
import collections, csv, glob, os, sys


def per_kernel(outdir, tag, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(outdir, tag, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc


def durations(outdir, tag):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(outdir, tag, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return acc


def main():
    outdir = sys.argv[1]
    fetch, write, dur = per_kernel(outdir, 'FETCH_SIZE', 'FETCH_SIZE'), per_kernel(outdir, 'WRITE_SIZE', 'WRITE_SIZE'), durations(outdir, 'FETCH_SIZE')
    rows = []
    for k in fetch:
        if k not in write or k not in dur:
            continue
        f, w, d = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k]), sum(dur[k]) / len(dur[k])
        rd, wr = 2.0 * f * 1024.0, w * 1024.0
        rows.append((sum(dur[k]), k, len(dur[k]), d, rd, wr))
    rows.sort(reverse=True)
    print('# HBM traffic per launch, every kernel of the default bench step (`tools/pmc_all_kernels.py`; PMC passes of `tools/pmc_traffic.sh`)\n')
    print('| kernel | launches | avg us (PMC pass) | read MB | written MB | TB/s |')
    print('|---|---|---|---|---|---|')
    for _, k, n, d, rd, wr in rows[:40]:
        print(f"| `{k[:90]}` | {n} | {d:.1f} | {rd / 1e6:.1f} | {wr / 1e6:.1f} | {(rd + wr) / d / 1e6:.2f} |")


if __name__ == '__main__':
    main()
