"""Shared helpers for the parity tests (golden loading, arch/params construction)."""
import os

import numpy as np
import torch

from iodine_amd import synth
from oracle import iodine_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

FAMILIES = {
    'tiny': lambda K, T: O.tiny_arch(slots=K, iters=T),
    'dsprites': lambda K, T: O.dsprites_arch(slots=K, iters=T),
    'clevr': lambda K, T: O.clevr_arch(slots=K, iters=T),
    # configs/test.yaml:26-52 and lib/config/defaults.py:35-100 (kernel sizes / ENCODING / SIGMA come from the fixture's meta entries)
    'testyaml': lambda K, T: O.Arch(dim_latent=16, iters=T, slots=K, img_size=64, ref_chan=32, ref_layers=3, ref_mlp=128, dec_chan=32, dec_layers=5),
    'defaults': lambda K, T: O.Arch(dim_latent=128, iters=T, slots=K, img_size=32, ref_chan=32, ref_layers=3, ref_mlp=256, dec_chan=64, dec_layers=5),
}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def golden_setup(g, dtype=torch.float32):
    """Rebuild (arch, params, x, eps, gt_masks) exactly as gen_goldens.py did."""
    K, T, B, S, L = (int(g[f'meta_{k}']) for k in 'KTBSL')
    fam, kind = str(g['meta_family']), str(g['meta_kind'])
    sw, sx, se = (int(v) for v in g['meta_seeds'])
    arch = FAMILIES[fam](K, T)
    if 'meta_encoding' in g.files:                      # cases generated with a non-default ARCH.ENCODING list
        arch.encoding = tuple(str(g['meta_encoding']).split(','))
    if 'meta_kernels' in g.files:                       # (REF.KERNEL_SIZE, DEC.KERNEL_SIZE) other than 3
        arch.ref_kernel, arch.dec_kernel = (int(v) for v in g['meta_kernels'])
    if 'meta_sigma' in g.files:
        arch.sigma = float(g['meta_sigma'])
    assert arch.img_size == S and arch.dim_latent == L
    shapes = O.param_shapes(arch)
    pn = synth.make_params(shapes, seed=sw, dec_gain=float(g['meta_dec_gain']),
                           posterior_scale=float(g['meta_post_scale']))
    params = {k: torch.from_numpy(v).to(dtype) for k, v in pn.items()}
    gt = None
    if kind == 'blobs':
        imgs, gt = synth.make_images(B, S, seed=sx, kind='blobs')
    else:
        imgs = synth.make_images(B, S, seed=sx, kind='uniform')
    eps = synth.make_eps(T, B, K, L, seed=se)
    return arch, params, torch.from_numpy(imgs).to(dtype), torch.from_numpy(eps).to(dtype), gt


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


# ---- HIP-side helpers (GPU tests) -------------------------------------------------------------
def hip_arch(arch):
    """oracle Arch -> ARCH-shaped namespace for iodine_amd.IODINE."""
    from iodine_amd.model import arch_namespace
    return arch_namespace(arch.dim_latent, arch.iters, arch.slots, arch.img_size,
                          (arch.ref_chan, arch.ref_layers, arch.ref_mlp), (arch.dec_chan, arch.dec_layers),
                          sigma=arch.sigma, layernorm=arch.layernorm, encoding=arch.encoding,
                          kernels=(arch.ref_kernel, arch.dec_kernel), ref_stride=arch.ref_stride)


def make_hip_model(arch, params, device='cuda:0', options=None):
    """options: library options set before the first call, e.g. {'conv_precision': 0} = the strict exact-fp32 path"""
    from iodine_amd import IODINE
    m = IODINE(hip_arch(arch))
    for k, v in (options or {}).items():
        m.set_option(k, v)
    sd = m.state_dict()
    assert list(sd.keys()) == list(params.keys()), 'state_dict names differ from the reference'
    m.load_state_dict({k: v.to(torch.float32) for k, v in params.items()})
    return m.to(device)


def nhwc(t):
    """(N,C,H,W) -> (N,H,W,C) contiguous"""
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def grad_views(name, got, ref):
    """The part of a parameter gradient that is compared PER TENSOR.  Everything - except element 3 of ``decoder.conv.bias``: the
    mask-logit bias has a mathematically ZERO gradient (the softmax over slots is invariant to a common offset, iodine.py:185), what
    is left there is rounding noise in the reference as well.  The three rgb bias gradients are well defined and ARE compared."""
    got, ref = np.asarray(got), np.asarray(ref)
    if name == 'decoder.conv.bias':
        return got[:3], ref[:3]
    return got, ref


def trajectory_setup(name, dtype=torch.float32):
    """(golden, arch, params, x, [eps of every step]) of a tests/golden/traj_*.npz fixture (gen_trajectory.py)."""
    tr = load_golden(name)
    g = load_golden(str(tr['meta_case']))
    arch, params, x, _, _ = golden_setup(g, dtype)
    K, T, B, L = (int(g[f'meta_{k}']) for k in 'KTBL')
    eps = [torch.from_numpy(synth.make_eps(T, B, K, L, seed=int(tr['meta_eps_seed0']) + s)).to(dtype)
           for s in range(int(tr['meta_steps']))]
    return tr, arch, params, x, eps


def check_trajectory_params(tr, tag, named_params, tol):
    """final parameters vs the reference's.  Adam moves every element by about lr per step whatever the size of its
    gradient, so the error is measured against the largest possible displacement lr * steps, not against |p|:
    full tensors (tiny) or the stored samples (larger cases), plus the sum of squares as a coarse whole-tensor check."""
    reach = float(tr['meta_lr']) * int(tr['meta_steps'])
    bad = []
    for n, p in named_params:
        a = p.detach().double().cpu()
        if n == 'decoder.conv.bias':
            # the mask-logit bias (element 3) has a mathematically ZERO gradient (softmax over slots is invariant to a
            # common offset, iodine.py:185): what reaches Adam is rounding noise, which it turns into +-lr steps - the
            # reference's own fp32 and fp64 runs end 0.9 * reach apart there.  Compare the three rgb biases only.
            ref3 = tr[f'{tag}.param.{n}'][:3] if f'{tag}.param.{n}' in tr.files else tr[f'{tag}.param.{n}.sample'][:3]
            e = float(np.abs(a.numpy()[:3] - ref3).max()) / reach
        elif f'{tag}.param.{n}' in tr.files:
            e = float(np.abs(a.numpy() - tr[f'{tag}.param.{n}']).max()) / reach
        else:
            flat = a.flatten()
            step = max(1, flat.numel() // 16)
            e = float(np.abs(flat[::step][:16].numpy() - tr[f'{tag}.param.{n}.sample']).max()) / reach
            ss, d = float(tr[f'{tag}.param.{n}.sumsq']), reach * tol * a.numel() ** 0.5     # ||p - p_ref|| <= d
            if abs(float((a * a).sum()) - ss) > 2.0 * ss ** 0.5 * d + d * d:
                e = float('inf')
        if not e < tol:
            bad.append((n, e))
    assert not bad, bad
