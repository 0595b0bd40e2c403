"""Kernel-level parity of the generic path's convolutions (kernels_generic.hip / kernels_gens2.hip) against PyTorch-CPU fp64 convs: the
stride-2 forms of the refinement stack with REF.KERNEL_SIZE 5 / 7 (reference: configs/test.yaml:40, lib/modeling/iodine.py:459,480) run
on v_mfma_f32_16x16x4_f32 since round 5 - forward, data gradient (four parity classes) and weight gradient, incl. the 17-of-20-channel
first layer, ragged image sizes and channel counts that are not multiples of 16."""
import pytest
import torch
import torch.nn.functional as F

from iodine_amd import _lib
from util import nhwc, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _op(mode, x, w, bias, aux, out, gb, n, si, ci, ldc, co, k, s, elu):
    L = _lib.lib()
    t = [v.to(DEV).contiguous() if v is not None else None for v in (x, w, bias, aux)]
    _lib.check(L.iodine_op_gen_conv(None, mode, _lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), _lib.ptr(t[3]), _lib.ptr(out),
                                    _lib.ptr(gb) if gb is not None else None, n, si, ci, ldc, co, k, s, elu), None, 'iodine_op_gen_conv')
    torch.cuda.synchronize()


CASES = [  # ci, ldc, co, k, S, N
    (17, 20, 32, 5, 64, 3), (32, 32, 32, 5, 32, 4), (32, 32, 32, 5, 16, 5), (17, 20, 64, 7, 32, 2), (64, 64, 64, 7, 16, 3),
    (17, 20, 32, 3, 32, 3), (24, 24, 24, 5, 36, 2), (48, 48, 48, 5, 18, 3), (8, 8, 8, 7, 20, 2), (17, 20, 128, 5, 32, 1),
    (32, 32, 32, 5, 7, 3), (32, 32, 32, 5, 64, 40)]


@pytest.mark.parametrize('ci,ldc,co,k,S,N', CASES)
def test_gen_stride2_forward(ci, ldc, co, k, S, N):
    x = _rand(N, ci, S, S, seed=50)
    w = _rand(co, ci, k, k, seed=51, scale=3.0 / (ci * k * k) ** 0.5)
    b = _rand(co, seed=52, scale=0.5)
    ref = nhwc(F.elu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=k // 2))).float()
    xp = torch.full((N, S, S, ldc), float('nan'))              # what lies past the ci real channels must never be read into the sum
    xp[..., :ci] = nhwc(x)
    out = torch.full(ref.shape, float('nan'), device=DEV)
    _op(0, xp, w, b, None, out, None, N, S, ci, ldc, co, k, 2, 1)
    assert rel_err(out.cpu(), ref) < 2e-6, rel_err(out.cpu(), ref)


@pytest.mark.parametrize('ci,ldc,co,k,S,N', [c for c in CASES if c[0] == c[1]])
def test_gen_stride2_dgrad(ci, ldc, co, k, S, N):
    So = (S - 1) // 2 + 1
    g = _rand(N, co, So, So, seed=53, scale=1e-2)
    w = _rand(co, ci, k, k, seed=54, scale=3.0 / (ci * k * k) ** 0.5)
    a = F.elu(_rand(N, ci, S, S, seed=55, scale=2.0))
    x = torch.zeros(N, ci, S, S, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x, w.double(), None, stride=2, padding=k // 2) * g.double()).sum().backward()
    ref = nhwc((x.grad * torch.where(a > 0, torch.ones_like(a), a + 1).double()).float())
    out = torch.full(ref.shape, float('nan'), device=DEV)
    _op(1, nhwc(g), w, None, nhwc(a), out, None, N, S, ci, ldc, co, k, 2, 0)
    assert rel_err(out.cpu(), ref) < 2e-6, rel_err(out.cpu(), ref)


@pytest.mark.parametrize('ci,ldc,co,k,S,N', CASES)
def test_gen_stride2_wgrad(ci, ldc, co, k, S, N):
    So = (S - 1) // 2 + 1
    x = _rand(N, ci, S, S, seed=56).double()
    d = _rand(N, co, So, So, seed=57, scale=1e-2).double()
    w = torch.zeros(co, ci, k, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x, w, b, stride=2, padding=k // 2) * d).sum().backward()
    xp = torch.full((N, S, S, ldc), float('nan'))
    xp[..., :ci] = nhwc(x.float())
    res = []
    for _ in range(2):
        gw, gb = torch.zeros(co, ci, k, k, device=DEV), torch.zeros(co, device=DEV)
        _op(2, xp, None, None, nhwc(d.float()), gw, gb, N, S, ci, ldc, co, k, 2, 0)
        res.append((gw.cpu(), gb.cpu()))
    assert rel_err(res[0][0], w.grad.float()) < 2e-6, rel_err(res[0][0], w.grad.float())
    assert rel_err(res[0][1], b.grad.float()) < 2e-6, rel_err(res[0][1], b.grad.float())
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])      # fixed summation order


def _random_cases(n, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    out = []
    for _ in range(n):
        ci = 4 * r(1, 18)
        out.append((ci, ci, 4 * r(1, 24), (3, 5, 7)[r(0, 2)], r(5, 40), r(1, 4)))
    return out


@pytest.mark.parametrize('ci,ldc,co,k,S,N', _random_cases(24, 7))
def test_gen_stride2_random_shapes(ci, ldc, co, k, S, N):
    """seeded sweep over channel counts (multiples of 4 up to 72 / 96), kernel sizes, ragged image sizes: all three directions of one layer.
    Gate 4e-6: a 7 x 7 conv over 64+ channels is a sum of > 3000 fp32 products accumulated four at a time - 2.4e-6 against fp64 is the
    accumulation order's rounding (the fp32 ATen conv of the reference sits at the same distance), not a wrong element."""
    So = (S - 1) // 2 + 1
    x = _rand(N, ci, S, S, seed=60).double()
    w = _rand(co, ci, k, k, seed=61, scale=3.0 / (ci * k * k) ** 0.5).double()
    b = _rand(co, seed=62, scale=0.5).double()
    d = _rand(N, co, So, So, seed=63, scale=1e-2).double()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, br, stride=2, padding=k // 2)
    (y * d).sum().backward()
    out = torch.full((N, So, So, co), float('nan'), device=DEV)
    _op(0, nhwc(x.float()), w.float(), b.float(), None, out, None, N, S, ci, ldc, co, k, 2, 1)
    e_f = rel_err(out.cpu(), nhwc(F.elu(y.detach())).float())
    assert e_f < 4e-6, e_f
    a = F.elu(_rand(N, ci, S, S, seed=64, scale=2.0))
    din = torch.full((N, S, S, ci), float('nan'), device=DEV)
    _op(1, nhwc(d.float()), w.float(), None, nhwc(a), din, None, N, S, ci, ldc, co, k, 2, 0)
    refd = nhwc((xr.grad * torch.where(a > 0, torch.ones_like(a), a + 1).double()).float())
    e_d = rel_err(din.cpu(), refd)
    assert e_d < 4e-6, e_d
    gw, gb = torch.zeros(co, ci, k, k, device=DEV), torch.zeros(co, device=DEV)
    _op(2, nhwc(x.float()), None, None, nhwc(d.float()), gw, gb, N, S, ci, ldc, co, k, 2, 0)
    e_w, e_b = rel_err(gw.cpu(), wr.grad.float()), rel_err(gb.cpu(), br.grad.float())
    assert e_w < 4e-6 and e_b < 4e-6, (e_w, e_b)


# ARCH.ENCODING subsets as internal-channel masks (code order of iodine.py:277-340: image 0-2, means 3-5, mask 6, mask_logits 7,
# mask_posterior 8, grad_means 9-11, grad_mask 12, likelihood 13, leave-one-out 14, coordinate 15-16): configs/test.yaml keeps
# {image, leave_one_out_likelihood}; the reference's default list drops the coordinates
_MASKS = {'test_yaml': 0b0000100000000000111, 'no_coordinate': 0x7fff, 'only_grads': 0b0001111000000000, 'full': 0x1ffff}


@pytest.mark.parametrize('name', sorted(_MASKS))
@pytest.mark.parametrize('k', [3, 5])
def test_gen_stride2_channel_mask_equals_the_plain_form(name, k):
    """The first refinement layer of an ENCODING subset: the library skips 4- / 16-channel groups whose channels are all absent (zero
    inputs x zero weights).  ADVICE r05: nothing compared the masked kernels with the unmasked ones - a wrong mask would silently drop real
    channels.  Inputs and weights of absent channels are zero, as pixel_pass2 / the weight expansion leave them; forward and weight
    gradient with the mask must be BITWISE what they are without it (skipped groups add exact zeros), and equal the fp64 conv."""
    mask = _MASKS[name]
    ci, ldc, co, S, N = 17, 20, 32, 32, 3
    on = torch.tensor([(mask >> c) & 1 for c in range(ci)], dtype=torch.float32)
    x = _rand(N, ci, S, S, seed=70) * on.view(1, ci, 1, 1)
    w = _rand(co, ci, k, k, seed=71, scale=3.0 / (ci * k * k) ** 0.5) * on.view(1, ci, 1, 1)
    b = _rand(co, seed=72, scale=0.5)
    So = (S - 1) // 2 + 1
    d = _rand(N, co, So, So, seed=73, scale=1e-2)
    ref = nhwc(F.elu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=k // 2))).float()
    xp = torch.zeros(N, S, S, ldc)
    xp[..., :ci] = nhwc(x)
    outs, grads = [], []
    for flag in (1, 1 | 0x100 | (mask << 9)):
        out = torch.full(ref.shape, float('nan'), device=DEV)
        _op(0, xp, w, b, None, out, None, N, S, ci, ldc, co, k, 2, flag)
        outs.append(out.cpu())
        gw, gb = torch.zeros(co, ci, k, k, device=DEV), torch.zeros(co, device=DEV)
        _op(2, xp, None, None, nhwc(d), gw, gb, N, S, ci, ldc, co, k, 2, flag & ~1)
        grads.append((gw.cpu(), gb.cpu()))
    assert rel_err(outs[0], ref) < 2e-6
    assert torch.equal(outs[0], outs[1])
    wd = torch.zeros(co, ci, k, k, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), wd, None, stride=2, padding=k // 2) * d.double()).sum().backward()
    # the skipped groups' weight gradients are exactly zero in both forms (their inputs are zero); live channels agree bitwise
    assert rel_err(grads[0][0], wd.grad.float()) < 2e-6
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
