#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of tools/pmc_traffic.sh into gpurun_out/<tag>_pmc.json (copied to profiles/): per-launch HBM bytes
(FETCH_SIZE x2 [gfx950: 16 B/lane reads are tallied at half, MI355X_MICROARCH.md "HBM"] + WRITE_SIZE, both reported in KB)
and matrix-pipe utilisation for the forward / data-gradient / weight-gradient launches of the decoder 3x3 conv, keyed by the
bench.py category names, with the digest of the kernel sources they were measured on (bench.py quotes the numbers only while
that digest matches the tree).

Since round 3 the HBM-bound helper kernels are recorded too (bytes only), keyed by bench.py's category names: bench.py's
``roofline_hbm.traffic`` comes from here.

    python tools/pmc_to_json.py gpurun_out/pmc_r03 r03 [--config dsprites ...]
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


HELPERS = (('dec_out_stream_f16x3_kernel', 'dec_out'), ('dec_out_rows_f16x3_kernel', 'dec_out'), ('dec_out_bwd_fused_f16x3_kernel', 'dec_out_bwd'), ('dec_l0_cells_kernel', 'dec_l0'),
           ('dec_out_dgrad_f16x3_kernel', 'dec_out_dgrad'), ('pixel_pass1_kernel', 'pixel_pass1'), ('pixel_pass2_kernel', 'pixel_pass2'),
           ('l0_rows_reduce_kernel', 'l0_reduce'), ('refine_head', 'refine_head'), ('conv3x3_s2_wgrad_f16x3_kernel', 'refine_wgrad'),
           ('refine_bwd01_kernel', 'refine_bwd01'), ('conv3x3_s2ws_f16x3_kernel', 'refine_conv'), ('refine_l0_fused_kernel', 'refine_l0f'),
           # kernels only the exact-fp32 path launches (rNN_pmc_strict.json)
           ('conv3x3_s2_wgrad_f32_kernel', 'refine_wgrad'), ('dec_out_wgrad_f32_kernel', 'dec_out_wgrad'), ('conv3x3_tile_kernel<4,', 'dec_out_dgrad'),
           ('dec_l0_kernel', 'dec_l0'))


def category(name):
    for frag, cat in HELPERS:
        if frag in name:
            return cat
    if 'conv3x3_s2_f16x3_kernel' in name:                    # <CR, CIN, COUT, MODE>: MODE 0 forward (CR 8 / 12 / 20 = first layer), 1 data gradient
        targs = name.split('<', 1)[1].split('>')[0].replace(' ', '').split(',')
        if targs[3] != '0':
            return 'refine_dgrad'
        return 'refine_l0' if targs[0] in ('8', '12', '20') else 'refine_conv'
    if 'conv3x3_wgrad_f16x3_ws_kernel' in name or 'conv3x3_wgrad_f16x3_kernel' in name or 'conv3x3_wgrad_f32_ws_kernel' in name:
        return 'conv_tile_wgrad'
    if 'conv3x3_ws_f16x3_kernel' in name:                    # <C, EPI>: 0 = forward, 1 / 4 = data gradient (stored / row sums)
        targs = name.split('<', 1)[1].split('>')[0].replace(' ', '').split(',')
        return 'conv_tile_fwd' if targs[1] == '0' else 'conv_tile_dgrad'
    if 'conv3x3_tile_f16x3_kernel' in name:
        targs = name.split('<', 1)[1].split('>')[0].replace(' ', '').split(',')
        if targs[0] != targs[1]:
            return None
        return 'conv_tile_fwd' if targs[2] == '0' else 'conv_tile_dgrad'
    return None


def per_launch(outdir, tag):
    files = glob.glob(os.path.join(outdir, tag, '**', '*counter_collection.csv'), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    names = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            c = category(r['Kernel_Name'])
            if c:
                acc[c][r['Counter_Name']].append(float(r['Counter_Value']))
                names[c] = r['Kernel_Name'][:120]
    return acc, names


def clocks(outdir, tag):
    """Shader clock while each kernel category runs: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration of the same
    dispatch (kernel trace of the same pass).  The chip clocks to its power budget (MI355X_MICROARCH.md "DVFS give-back"), so
    matrix-pipe utilisation has to be read together with this number."""
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(outdir, tag, '**', '*counter_collection.csv'), recursive=True):
        kt = f.replace('counter_collection', 'kernel_trace')
        if not os.path.exists(kt):
            continue
        dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(kt))}
        for r in csv.DictReader(open(f)):
            c = category(r['Kernel_Name'])
            if c and r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and dur.get(r['Dispatch_Id'], 0) > 0:
                acc[c].append((float(r['Counter_Value']) / 8.0 / dur[r['Dispatch_Id']], dur[r['Dispatch_Id']] / 1e3))
    return acc


def main():
    outdir = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else 'r05'
    cfg = 'dsprites' if 'dsprites' in sys.argv[2:] else 'clevr6'
    strict = '0' in [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == '--conv-precision']      # passes over the exact-fp32 path
    from iodine_amd.build import source_digest
    # the GPU box has no .git: the commit is handed over in the environment (tools/round_profiles.sh, IODINE_COMMIT=$(git rev-parse HEAD))
    commit = os.environ.get('IODINE_COMMIT', '').strip()
    if not commit:
        try:
            commit = subprocess.run(['git', 'rev-parse', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip() or 'unknown'
        except Exception:
            commit = 'unknown'
    fetch, names = per_launch(outdir, 'FETCH_SIZE')
    write, _ = per_launch(outdir, 'WRITE_SIZE')
    mfma, _ = per_launch(outdir, 'SQ_VALU_MFMA_BUSY_CYCLES')
    clk = clocks(outdir, 'SQ_VALU_MFMA_BUSY_CYCLES')
    kernels = {}
    cats = ('conv_tile_fwd', 'conv_tile_dgrad', 'conv_tile_wgrad') + tuple(sorted(set(fetch) - {'conv_tile_fwd', 'conv_tile_dgrad', 'conv_tile_wgrad'}))
    for c in cats:
        if c not in fetch or c not in write:
            continue
        f = fetch[c]['FETCH_SIZE']
        w = write[c]['WRITE_SIZE']
        fkb, wkb = sum(f) / len(f), sum(w) / len(w)
        k = dict(kernel=names[c], launches_sampled=len(f), fetch_size_kb_raw=round(fkb, 1), write_size_kb=round(wkb, 1),
                 hbm_bytes_per_launch=round((2.0 * fkb + wkb) * 1024.0))
        if c in mfma and mfma[c].get('GRBM_GUI_ACTIVE'):
            busy = sum(mfma[c]['SQ_VALU_MFMA_BUSY_CYCLES']) / len(mfma[c]['SQ_VALU_MFMA_BUSY_CYCLES'])
            gui = sum(mfma[c]['GRBM_GUI_ACTIVE']) / len(mfma[c]['GRBM_GUI_ACTIVE'])
            k['mfma_busy_cycles'] = round(busy)
            k['grbm_gui_active'] = round(gui)
            k['mfma_util'] = round(busy / (gui / 8.0 * 1024.0), 4)        # busy cycles / (cycles per XCD x 1024 SIMDs)
            if clk.get(c):
                k['clock_ghz'] = round(sum(v[0] for v in clk[c]) / len(clk[c]), 3)
                k['us_in_pmc_pass'] = round(sum(v[1] for v in clk[c]) / len(clk[c]), 1)
                # fraction of the 2.4 GHz matrix peak the launch sustains: utilisation x clock / 2.4
                k['mfma_rate_of_2p4ghz_peak'] = round(k['mfma_util'] * k['clock_ghz'] / 2.4, 4)
        kernels[c] = k
    rec = dict(commit=commit, csrc_sha256=source_digest(), conv_precision=0 if strict else 1,
               shape=dict(config=cfg, batch=32, slots=7 if cfg == 'clevr6' else 6),
               method='rocprofv3 --pmc <one set per run> --kernel-trace over bench.py --steps 2 --warmup 1; '
                      'hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (tools/pmc_traffic.sh)',
               kernels=kernels)
    path = os.path.join(ROOT, 'gpurun_out', f'{tag}_pmc_strict.json' if strict else f'{tag}_pmc.json')
    json.dump(rec, open(path, 'w'), indent=1)
    print(json.dumps(rec, indent=1))
    print('wrote', path, '(copy to profiles/)')


if __name__ == '__main__':
    main()
