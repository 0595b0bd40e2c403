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
}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def golden_setup(g, dtype=torch.float32):
    """Rebuild (arch, params, x, eps, gt_masks) exactly as gen_goldens.py did."""
    K, T, B, S, L = (int(g[f'meta_{k}']) for k in 'KTBSL')
    fam, kind = str(g['meta_family']), str(g['meta_kind'])
    sw, sx, se = (int(v) for v in g['meta_seeds'])
    arch = FAMILIES[fam](K, T)
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
                          sigma=arch.sigma, layernorm=arch.layernorm)


def make_hip_model(arch, params, device='cuda:0'):
    from iodine_amd import IODINE
    m = IODINE(hip_arch(arch))
    sd = m.state_dict()
    assert list(sd.keys()) == list(params.keys()), 'state_dict names differ from the reference'
    m.load_state_dict({k: v.to(torch.float32) for k, v in params.items()})
    return m.to(device)


def nhwc(t):
    """(N,C,H,W) -> (N,H,W,C) contiguous"""
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()
