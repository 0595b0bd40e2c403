#!/usr/bin/env python3
"""CPU simulation of split-precision convolutions through the whole IODINE training step.

Every conv of the oracle is replaced by  conv(a_hi, w_hi) + conv(a_hi, w_lo) + conv(a_lo, w_hi)  with hi = round16(v),
lo = round16(v - hi) (products of 16-bit values are exact in fp32, accumulation is fp32), and the resulting loss / ELBOs /
parameter gradients are compared with the reference's fp64 goldens.  This is the experiment that justified the 3 x fp16 MFMA
decoder kernels.  Test infrastructure: uses oracle/ and tests/golden.

    python tools/split_precision_sim.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch
import torch.nn.functional as F

from oracle import iodine_oracle as O
from util import golden_setup, load_golden

torch.set_num_threads(8)
_conv = F.conv2d
MODE = {'m': 'fp32'}


def _split(t, dt):
    hi = t.to(dt).to(torch.float32)
    return hi, (t - hi).to(dt).to(torch.float32)


def conv_sim(x, w, b=None, stride=1, padding=0):
    m = MODE['m']
    if m == 'fp32' or x.dtype != torch.float32:
        return _conv(x, w, b, stride=stride, padding=padding)
    dt = torch.bfloat16 if 'bf16' in m else torch.float16
    if m.endswith('x1'):
        return _conv(x.to(dt).float(), w.to(dt).float(), b, stride=stride, padding=padding)
    xh, xl = _split(x, dt)
    wh, wl = _split(w, dt)
    return (_conv(xh, wh, b, stride=stride, padding=padding) + _conv(xh, wl, None, stride=stride, padding=padding)
            + _conv(xl, wh, None, stride=stride, padding=padding))


def main():
    O.F.conv2d = conv_sim
    for case in ('cfg1_dsprites_k4_t3_b4', 'cfg2_dsprites_k6_t5_b2'):
        g = load_golden(case)
        arch, params, x, eps, _ = golden_setup(g)
        ref_loss, ref_el = float(g['f64.train.loss']), g['f64.train.elbos']
        for m in ('fp32', 'bf16x1', 'fp16x1', 'bf16x3', 'fp16x3'):
            MODE['m'] = m
            out, grads = O.train_step_grads(x, eps, params, arch)
            el = out['elbos'].double().numpy()
            gerr = []
            for n, gv in grads.items():
                a = gv.double().flatten()
                ref = float(g[f'f64.train.grad.{n}.sumsq'])
                step = max(1, a.numel() // 16)
                gerr.append(np.abs(a[::step][:16].numpy() - g[f'f64.train.grad.{n}.sample']).max() / np.sqrt(ref / a.numel()))
            print(f'{case} {m:7s} loss_rel={abs(out["loss"].item() - ref_loss) / abs(ref_loss):.2e} '
                  f'elbo_rel_max={np.abs((el - ref_el) / ref_el).max():.2e} grad_err/rms max={max(gerr):.2e}')
    O.F.conv2d = _conv


if __name__ == '__main__':
    main()
