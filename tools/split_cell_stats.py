#!/usr/bin/env python3
"""Where do the operands of the split-fp16 convs sit relative to their cell scale?  (VERDICT r03, next #3b; CPU only, test infrastructure:
uses the oracle.)

The decoder's 3x3 convs take fp32 tensors and split every element on the fly into fp16 hi + lo at ONE power-of-two scale per 8 x 16 pixel
cell (all channels; the cell max is mapped into [2^12, 2^13)).  An element at r = |x| / cell max keeps 22 bits while its lo part is a
normal fp16 number, i.e. for r >= 2^-15; below that lo is subnormal (absolute step 2^-24 at the cell's scale) and the element's relative
error grows as 2^-37 / r (r = 2^-20: 2^-17, r = 2^-24: 2^-13).  fp32 would keep 2^-24 everywhere.  This script measures, for the tensors
the BACKWARD convs read (d loss / d pre-activation of every decoder layer, every refinement iteration - the tensors with the widest dynamic
range: r (x - mu) / sigma^2 under a saturated softmax), on trained and artificially sharpened checkpoints:

  * the histogram of log2 r over all elements,
  * per OUTPUT element of the consuming data-gradient conv: the share of its magnitude sum_k |w_k| |x_k| that comes from input elements
    with r < 2^-10 and with r < 2^-15,
  * the resulting bound on the output's error, sum |w| |x| e(r) with e(r) = max(2^-22, 2^-37 / r) (+ 2^-22 for the weights' own split),
    (a) relative to the output's own magnitude sum |w| |x| (meaningless where the output is itself far below its surroundings) and
    (b) relative to the LARGEST such magnitude in the 3 x 3 cells around the output - the scale every consumer of the tensor works at
    (the next conv takes its tile scale from exactly that neighbourhood; weight gradients, class sums and the ELBO add the
    neighbourhood's small and large values up in fp32, where anything below 2^-24 of the running sum is lost in the reference too).

usage: python tools/split_cell_stats.py [fixture:ckpt:sharpen ...]     (default: the trained-weights cases of tests/test_gpu_trained_weights.py)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from iodine_amd import synth  # noqa: E402
from oracle import iodine_oracle as O  # noqa: E402

CELL_H, CELL_W = 8, 16
BINS = [0, -2, -4, -6, -8, -10, -12, -15, -18, -21, -24, -1000]          # log2 r bin edges


def cell_ratio(t):
    """t (N, C, S, S) -> r = |t| / max |t| over the element's 8 x 16 cell (all channels of a slot-image); cells with max 0 give r = 1"""
    N, C, S, _ = t.shape
    a = t.abs()
    ch, cw = min(CELL_H, S), min(CELL_W, S)
    m = a.reshape(N, C, S // ch, ch, S // cw, cw).amax(dim=(1, 3, 5), keepdim=True)
    m = m.expand(N, C, S // ch, ch, S // cw, cw).reshape(N, C, S, S)
    return torch.where(m > 0, a / m, torch.ones_like(a))


def elem_err(r):
    """relative error bound of one split element at cell ratio r (see the module docstring)"""
    return torch.clamp(2.0 ** -37 / torch.clamp(r, min=2.0 ** -60), min=2.0 ** -22)


def analyse(name, ckpt, sharpen):
    t = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    fam, K, T, B = str(t['meta_family']), int(t['meta_K']), int(t['meta_T']), int(t['meta_B'])
    arch = {'tiny': O.tiny_arch, 'dsprites': O.dsprites_arch, 'clevr': O.clevr_arch}[fam](slots=K, iters=T)
    sw, sx, se = (int(v) for v in t['meta_seeds'])
    p = {k: torch.from_numpy(t[f'ckpt{ckpt}.param.{k}']).clone().double() for k in O.param_shapes(arch)}
    if sharpen != 1.0:
        p['decoder.conv.weight'][3] *= sharpen
        p['decoder.conv.bias'][3] *= sharpen
    imgs, _ = synth.make_images(B, arch.img_size, seed=sx, kind='blobs')
    x = torch.from_numpy(imgs).double()
    eps = torch.from_numpy(synth.make_eps(T, B, K, arch.dim_latent, seed=se + 5000 + ckpt)).double()
    # the refinement trajectory (posterior before every elbo() call) from the oracle's own loop; then each decoder pass is re-run with the
    # pre-activations exposed (the oracle keeps only activations)
    trace = []
    O._loop(x, eps, p, arch, False, trace)
    post = [(tr['post_mean'].detach(), tr['post_logvar'].detach()) for tr in trace]
    hist = np.zeros(len(BINS) - 1)
    n_elem = 0
    share10, share15, bound, bound_cell = [], [], [], []
    sharp = []
    for i, (pm, plv) in enumerate(post):
        z = O.sample(pm, plv, eps[i])
        h = O.spatial_broadcast(z.reshape(B * K, -1), arch.img_size).clone().requires_grad_(True)
        pres = []
        for l in range(arch.dec_layers):
            pre = F.conv2d(h, p[f'decoder.mlc.layers.{l}.weight'], p[f'decoder.mlc.layers.{l}.bias'], padding=arch.dec_kernel // 2)
            pre.retain_grad()
            pres.append(pre)
            h = F.elu(pre)
        out = F.conv2d(h, p['decoder.conv.weight'], p['decoder.conv.bias'], padding=arch.dec_kernel // 2)
        S = arch.img_size
        rgb, logit = torch.split(out, [3, 1], dim=1)
        mean, logits = torch.sigmoid(rgb).reshape(B, K, 3, S, S), logit.reshape(B, K, 1, S, S)
        mask = F.softmax(logits, dim=1)
        ll = torch.logsumexp(torch.log(mask + 1e-12) + O.gaussian_log_likelihood(x[:, None], mean, arch.sigma), dim=1)
        (ll.sum()).backward()                                   # = d (B * ELBO) / d . up to the KL term, which does not reach the decoder
        sharp.append(float(mask.detach().max(dim=1).values.mean()))
        # layer l's data gradient reads dpre[l] (l = D-1 .. 1) with the transposed weights of layer l
        for l in range(arch.dec_layers - 1, 0, -1):
            d = pres[l].grad.detach()
            r = cell_ratio(d)
            lr = torch.log2(torch.clamp(r, min=2.0 ** -999)).flatten().numpy()
            hist += np.histogram(-lr, bins=[-b for b in BINS])[0]
            n_elem += lr.size
            w = p[f'decoder.mlc.layers.{l}.weight'].abs()
            a = d.abs()
            tot = F.conv_transpose2d(a, w, padding=arch.dec_kernel // 2)
            s10 = F.conv_transpose2d(a * (r < 2.0 ** -10), w, padding=arch.dec_kernel // 2)
            s15 = F.conv_transpose2d(a * (r < 2.0 ** -15), w, padding=arch.dec_kernel // 2)
            eb = F.conv_transpose2d(a * elem_err(r), w, padding=arch.dec_kernel // 2)
            ok = tot > 0
            share10.append((s10[ok] / tot[ok]).flatten())
            share15.append((s15[ok] / tot[ok]).flatten())
            bound.append((eb[ok] / tot[ok] + 2.0 ** -22).flatten())
            N_, C_, S_, _ = tot.shape
            ch, cw = min(CELL_H, S_), min(CELL_W, S_)
            tmax = tot.reshape(N_, C_, S_ // ch, ch, S_ // cw, cw).amax(dim=(1, 3, 5))                 # (N, cells y, cells x)
            tmax = F.max_pool2d(tmax[:, None], 3, stride=1, padding=1)[:, 0]                          # 3 x 3 cells around: the kernel's tile scale
            tmax = tmax[:, None, :, None, :, None].expand(N_, C_, S_ // ch, ch, S_ // cw, cw).reshape(N_, C_, S_, S_)
            okc = tmax > 0
            bound_cell.append((eb[okc] / tmax[okc] + 2.0 ** -22).flatten())
    s10, s15, bd, bc = torch.cat(share10), torch.cat(share15), torch.cat(bound), torch.cat(bound_cell)

    def q(v, f):
        return float(torch.quantile(v[:: max(1, v.numel() // 2_000_000)], f))
    row = dict(case=f'{name} ckpt {ckpt} x{sharpen:g}', sharp=float(np.mean(sharp[-1:])), hist=hist / n_elem,
               s10=(q(s10, 0.5), q(s10, 0.999), float(s10.max())), s15=(q(s15, 0.5), q(s15, 0.999), float(s15.max())),
               bound=(q(bd, 0.5), q(bd, 0.999), float(bd.max())), bound_cell=(q(bc, 0.5), q(bc, 0.999), float(bc.max())))
    return row


def main():
    torch.set_num_threads(int(os.environ.get('STATS_THREADS', '8')))
    cases = [c.split(':') for c in sys.argv[1:]] or [('teacher_cfg1_long', 3000, 1), ('teacher_cfg1_long', 3000, 8), ('teacher_cfg1_long', 3000, 32),
                                                     ('teacher_cfg1_long', 3000, 128), ('teacher_cfg3', 12, 1), ('teacher_cfg3', 12, 8),
                                                     ('teacher_cfg3', 12, 32)]
    rows = [analyse(c[0], int(c[1]), float(c[2])) for c in cases]
    edges = [f'[2^{BINS[i + 1]}, 2^{BINS[i]})' if BINS[i + 1] > -999 else f'< 2^{BINS[i]}' for i in range(len(BINS) - 1)]
    print('| case | mean max-mask (last iteration) | ' + ' | '.join(edges) + ' |')
    print('|---|---|' + '---|' * len(edges))
    for r in rows:
        print(f"| {r['case']} | {r['sharp']:.3f} | " + ' | '.join(f'{v:.2e}' if 0 < v < 1e-3 else f'{v:.4f}' for v in r['hist']) + ' |')
    print()
    print('| case | share of an output from r < 2^-10: median / 99.9 % / max | from r < 2^-15: median / 99.9 % / max | error bound relative to '
          'the output\'s own sum of magnitudes: median / 99.9 % / max | error bound relative to the largest sum of magnitudes in the 3 x 3 cells around the output: median / 99.9 % / max |')
    print('|---|---|---|---|---|')
    for r in rows:
        print(f"| {r['case']} | " + ' / '.join(f'{v:.1e}' for v in r['s10']) + ' | ' + ' / '.join(f'{v:.1e}' for v in r['s15']) + ' | '
              + ' / '.join(f'{v:.1e}' for v in r['bound']) + ' | ' + ' / '.join(f'{v:.1e}' for v in r['bound_cell']) + ' |')


if __name__ == '__main__':
    main()
