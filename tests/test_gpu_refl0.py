"""Fused encoding + first refinement layer (kernels_refl0.hip; get_input_encoding, lib/modeling/iodine.py:243-343, followed by
RefinementNetwork.mlc.layers[0], iodine.py:459,480) against the CPU oracle and against the three-kernel form it replaces
(pixel_pass2 + two stride-2 convs, option refine_l0_fused 0): image borders (zero padding of the stride-2 conv across tile edges),
slot counts 1 / 3 / 9 (9 = the largest the kernel's LDS holds), ENCODING subsets (channel mask), one-tile and many-tile images."""
import pytest
import torch

from iodine_amd import synth
from oracle import iodine_oracle as O
from util import grad_views, make_hip_model, rel_err, rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _case(arch, B, seed):
    pn = synth.make_params(O.param_shapes(arch), seed=seed, dec_gain=3.0, posterior_scale=0.05)
    params = {k: torch.from_numpy(v) for k, v in pn.items()}
    imgs, _ = synth.make_images(B, arch.img_size, seed=seed + 1, kind='blobs')
    eps = torch.from_numpy(synth.make_eps(arch.iters, B, arch.slots, arch.dim_latent, seed=seed + 2))
    return params, torch.from_numpy(imgs), eps


def _arch(slots, iters, img, encoding=None):
    a = O.tiny_arch(slots=slots, iters=iters, img_size=img, chan=64, mlp=32, ref_layers=2, dec_layers=2)
    if encoding is not None:
        a.encoding = encoding
    return a


CASES = {
    'k3_32px_b2': (_arch(3, 2, 32), 2),                       # one tile column, eight tile rows
    'k1_64px_b1': (_arch(1, 2, 64), 1),
    'k9_64px_b1': (_arch(9, 1, 64), 1),                       # 158 KB of LDS
    'k4_128px_b1': (_arch(4, 1, 128), 1),                     # 4 x 32 tiles: interior tiles and all four borders
    'k3_64px_default_encoding': (_arch(3, 2, 64, tuple(e for e in O.FULL_ENCODING if e != 'coordinate')), 2),
    'k2_32px_subset': (_arch(2, 2, 32, tuple(e for e in O.FULL_ENCODING if e not in ('mask_posterior', 'grad_mask', 'image'))), 3),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_fused_first_layer_against_oracle_and_three_kernel_form(name):
    arch, B = CASES[name]
    params, x, eps = _case(arch, B, seed=300 + sorted(CASES).index(name))
    xd, ed = x.to(DEV), eps.to(DEV)
    ref = O.reconstruct(x, eps, params, arch)
    out, rg = O.train_step_grads(x, eps, params, arch)
    res = {}
    for fused in (1, 0):
        m = make_hip_model(arch, params)
        m.set_option('refine_l0_fused', fused)
        # stage: encoding and the layer's output after the first iteration (debug runs materialise the encoding in both forms)
        m.set_option('stop_after_iters', 1)
        m.reconstruct(xd, ed)
        enc = m.debug_buffer('enc').cpu().clone()
        act = m.debug_buffer('ract0', 0).cpu().clone()
        m.set_option('stop_after_iters', -1)
        pred, mask, mean = m.reconstruct(xd, ed)
        assert rel_err(m.elbo_terms.cpu()[:, 0], ref['elbos']) < 1e-4, fused
        assert rel_err(pred.cpu(), ref['pred']) < 2e-4 and rel_err(mask.cpu(), ref['mask']) < 2e-4, fused
        p2, k2, _ = m.reconstruct(xd, ed)
        assert torch.equal(p2, pred) and torch.equal(k2, mask)              # deterministic
        m.zero_grad(set_to_none=True)
        loss = m(xd, ed)
        loss.backward()
        assert abs(loss.item() - float(out['loss'])) <= 1e-5 * abs(float(out['loss'])), fused
        bad = [(n, rel_l2(*grad_views(n, p.grad.cpu().numpy(), rg[n].numpy()))) for n, p in m.named_parameters()
               if not rel_l2(*grad_views(n, p.grad.cpu().numpy(), rg[n].numpy())) < 1e-3]
        assert not bad, (fused, bad)
        res[fused] = (enc, act, pred.cpu(), {n: p.grad.cpu().clone() for n, p in m.named_parameters()})
    # the encoding is the same function of the same inputs in both forms (one definition of the per-pixel terms): bitwise
    assert torch.equal(res[1][0], res[0][0])
    # the layer output differs only by the summation order / the scale granularity of the fp16 split
    assert rel_err(res[1][1], res[0][1]) < 2e-6
    assert rel_err(res[1][2], res[0][2]) < 1e-4
    for n in res[0][3]:
        assert rel_l2(res[1][3][n].numpy(), res[0][3][n].numpy()) < 1e-4, n
