#!/usr/bin/env python3
"""Time the split-fp16 tile conv kernels (op-level) at the cfg3 shape, interleaving ablation variants.
usage: conv_ablate.py <mode> <flags,flags,...>   (IODINE_CONV_ABLATE is read once per process, so each variant
runs in a child process; rounds are interleaved to average out clock drift).
Flags understood by conv_variant=3 (mode 4): (1 = skip the MFMAs: gone since the fragment reads are interleaved with them), 4 skip the weight LDS-DMA, 8 fixed scale (no max), 16 skip the consumers'
epilogue, 32 skip the input staging writes.  Results are WRONG with any flag set: timing only."""
import os, subprocess, sys, time
if len(sys.argv) > 2 and sys.argv[2] != 'child':
    mode = sys.argv[1]; variants = sys.argv[2].split(',')
    res = {v: [] for v in variants}
    for rnd in range(3):
        for v in variants:
            env = dict(os.environ, IODINE_CONV_ABLATE=v)
            o = subprocess.run([sys.executable, __file__, mode, 'child'], env=env, capture_output=True, text=True).stdout
            res[v].append(float(o.strip().split()[-1]))
    for v in variants:
        print(f'mode {mode} flags {v:>3}: ' + ' '.join(f'{t:.3f}' for t in res[v]) + f'  min {min(res[v]):.3f} ms')
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iodine_amd import _lib
L = _lib.lib()
mode = int(sys.argv[1])
N, S, C = 224, 128, 64
x = torch.rand(N, S, S, C, device='cuda') * 2 - 1
w = (torch.rand(C, C, 3, 3, device='cuda') * 2 - 1) * 0.1
b = torch.zeros(C, device='cuda')
out = torch.empty_like(x)
def run():
    _lib.check(L.iodine_op_conv3x3(None, mode, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(out), N, S, S, C, C, C, C, 1, 0, 0))
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(min(ts) * 1e3)
