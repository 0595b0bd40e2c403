"""CPU restatement of the IODINE iterative-refinement step (test infrastructure).

Every function cites the reference lines it follows (paths are relative to
``/root/reference``).  The arithmetic is plain PyTorch-CPU (the same ATen ops
the reference itself runs on its CPU path), written functionally over a flat
``{state_dict name: tensor}`` parameter dictionary with an EXPLICIT epsilon
stream ``eps[T+1, B, K, L]`` instead of ``torch.randn_like`` so that the HIP
path and this oracle can be driven with identical noise.

Quirks of the reference that are reproduced on purpose:
  * KL is taken against N(0, 1), not the learned initial posterior
    (lib/modeling/iodine.py:653-659).
  * ``log(mask + 1e-12)`` inside the mixture (iodine.py:213-216).
  * mask_posterior / leave-one-out use un-stabilised ``exp`` (iodine.py:286-293,317-331).
  * 5-D layer-norm uses the biased std, 3-D layer-norm ``torch.std`` (unbiased);
    both divide by ``std + 1e-5`` (iodine.py:376-395).
  * The MLP applies ELU and the caller applies it again (iodine.py:485,565).
  * ``(c, h) = lstm(x, hidden)`` swaps the names, so the posterior update is read
    out of the CELL state c1, while the state handed to the next iteration keeps
    torch's (h1, c1) order (iodine.py:488-503).
  * Refinement inputs are detached (iodine.py:343); lambda is detached before the
    additive update (iodine.py:642-643) so the outer backward reaches earlier
    iterations only through the LSTM state.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

FULL_ENCODING = (
    'posterior', 'grad_post', 'image', 'means', 'mask', 'mask_logits',
    'mask_posterior', 'grad_means', 'grad_mask', 'likelihood',
    'leave_one_out_likelihood', 'coordinate',
)
# the reference's default ARCH.ENCODING (lib/config/defaults.py:57-80): everything but 'coordinate'
DEFAULT_ENCODING = tuple(e for e in FULL_ENCODING if e != 'coordinate')


@dataclass
class Arch:
    """Mirror of the ``ARCH.*`` config node read by iodine.py:8-32."""
    dim_latent: int = 64
    iters: int = 5
    slots: int = 7
    sigma: float = 0.10
    img_size: int = 128
    img_channels: int = 3
    layernorm: bool = True
    ref_chan: int = 64
    ref_layers: int = 4
    ref_mlp: int = 256
    ref_kernel: int = 3
    ref_stride: int = 2
    dec_chan: int = 64
    dec_layers: int = 4
    dec_kernel: int = 3
    encoding: Tuple[str, ...] = field(default_factory=lambda: FULL_ENCODING)

    @property
    def n_input_channels(self) -> int:
        # iodine.py:345-374; the full encoding list gives 3+3+1+1+1+3+1+1+1+2 = 17, the reference's DEFAULT list
        # (lib/config/defaults.py:57-80, no 'coordinate') 15
        c = self.img_channels
        sizes = dict(image=c, means=c, mask=1, mask_logits=1, mask_posterior=1, grad_means=c, grad_mask=1, likelihood=1,
                     leave_one_out_likelihood=1, coordinate=2)
        return sum(n for k, n in sizes.items() if k in self.encoding)

    @property
    def latent_size(self) -> int:
        # iodine.py:345-374: 2L for 'posterior' + 2L for 'grad_post'
        return 2 * self.dim_latent * (('posterior' in self.encoding) + ('grad_post' in self.encoding))


def clevr_arch(slots=7, iters=5) -> Arch:
    """configs/clevr6_prop.yaml:26-45."""
    return Arch(dim_latent=64, iters=iters, slots=slots, sigma=0.10, img_size=128,
                ref_chan=64, ref_layers=4, ref_mlp=256, dec_chan=64, dec_layers=4)


def dsprites_arch(slots=6, iters=5) -> Arch:
    """configs/dsprites_noclip.yaml:26-45."""
    return Arch(dim_latent=16, iters=iters, slots=slots, sigma=0.10, img_size=64,
                ref_chan=32, ref_layers=3, ref_mlp=128, dec_chan=32, dec_layers=5)


def tiny_arch(slots=3, iters=2, img_size=16, dim_latent=8, chan=32, mlp=32,
              ref_layers=2, dec_layers=2) -> Arch:
    """kB-sized architecture for per-op fixtures."""
    return Arch(dim_latent=dim_latent, iters=iters, slots=slots, sigma=0.10,
                img_size=img_size, ref_chan=chan, ref_layers=ref_layers, ref_mlp=mlp,
                dec_chan=chan, dec_layers=dec_layers)


def param_shapes(a: Arch) -> "Dict[str, Tuple[int, ...]]":
    """state_dict names/shapes of the reference module tree (iodine.py:26-33,
    412-423, 446-464, 543-557, 570-584, 596-604), in ``named_parameters`` order."""
    L, H = a.dim_latent, a.ref_mlp
    shapes: Dict[str, Tuple[int, ...]] = {}
    cin = a.n_input_channels
    for i in range(a.ref_layers):
        shapes[f'refine.mlc.layers.{i}.weight'] = (a.ref_chan, cin, a.ref_kernel, a.ref_kernel)
        shapes[f'refine.mlc.layers.{i}.bias'] = (a.ref_chan,)
        cin = a.ref_chan
    shapes['refine.mlp.layers.0.weight'] = (H, a.ref_chan)
    shapes['refine.mlp.layers.0.bias'] = (H,)
    shapes['refine.lstm.weight_ih'] = (4 * H, H + 4 * L)      # iodine.py:462 hard-codes 4 L: both latent entries must be enabled
    shapes['refine.lstm.weight_hh'] = (4 * H, H)
    shapes['refine.lstm.bias_ih'] = (4 * H,)
    shapes['refine.lstm.bias_hh'] = (4 * H,)
    shapes['refine.mean_update.weight'] = (L, H)
    shapes['refine.mean_update.bias'] = (L,)
    shapes['refine.logvar_update.weight'] = (L, H)
    shapes['refine.logvar_update.bias'] = (L,)
    cin = L + 2
    for i in range(a.dec_layers):
        shapes[f'decoder.mlc.layers.{i}.weight'] = (a.dec_chan, cin, a.dec_kernel, a.dec_kernel)
        shapes[f'decoder.mlc.layers.{i}.bias'] = (a.dec_chan,)
        cin = a.dec_chan
    shapes['decoder.conv.weight'] = (4, a.dec_chan, a.dec_kernel, a.dec_kernel)
    shapes['decoder.conv.bias'] = (4,)
    shapes['posterior.init_mean'] = (L,)
    shapes['posterior.init_logvar'] = (L,)
    return shapes


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------

def coord_planes(S: int, dtype, device=None) -> Tensor:
    """(2, S, S): plane 0 = x (varies along W), plane 1 = y (varies along H).
    iodine.py:526-530 and 334-338 (``linspace(-1, 1, S)``, default meshgrid 'ij')."""
    lin = torch.linspace(-1, 1, S, device=device).to(dtype)
    yy = lin[:, None].expand(S, S)
    xx = lin[None, :].expand(S, S)
    return torch.stack((xx, yy), dim=0)


def spatial_broadcast(z: Tensor, S: int) -> Tensor:
    """(N, L) -> (N, L+2, S, S).  iodine.py:512-540."""
    N, L = z.shape
    tiled = z[:, :, None, None].expand(N, L, S, S)
    coords = coord_planes(S, z.dtype, z.device)[None].expand(N, 2, S, S)
    return torch.cat((tiled, coords), dim=1)


def conv_stack(x: Tensor, p: Dict[str, Tensor], prefix: str, n_layers: int,
               kernel: int, stride: int, keep: Optional[List[Tensor]] = None) -> Tensor:
    """``MultiLayerConv``: n x [conv k x k, pad k//2, stride] + ELU.  iodine.py:570-594."""
    for i in range(n_layers):
        x = F.conv2d(x, p[f'{prefix}.layers.{i}.weight'], p[f'{prefix}.layers.{i}.bias'],
                     stride=stride, padding=kernel // 2)
        x = F.elu(x)
        if keep is not None:
            keep.append(x)
    return x


def decoder(z: Tensor, p: Dict[str, Tensor], a: Arch,
            keep: Optional[List[Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """z (B, K, L) -> mean (B, K, 3, S, S) in (0,1), mask logits (B, K, 1, S, S).
    iodine.py:425-444."""
    B, K, L = z.shape
    h = spatial_broadcast(z.reshape(B * K, L), a.img_size)
    h = conv_stack(h, p, 'decoder.mlc', a.dec_layers, a.dec_kernel, 1, keep)
    out = F.conv2d(h, p['decoder.conv.weight'], p['decoder.conv.bias'],
                   stride=1, padding=a.dec_kernel // 2)
    rgb, logit = torch.split(out, [3, 1], dim=1)
    mean = torch.sigmoid(rgb)
    S = a.img_size
    return mean.reshape(B, K, 3, S, S), logit.reshape(B, K, 1, S, S)


def gaussian_log_likelihood(x: Tensor, loc: Tensor, scale: float) -> Tensor:
    """iodine.py:661-666."""
    return -(x - loc) ** 2 / (2 * scale ** 2) - math.log(scale) - 0.5 * math.log(2 * math.pi)


def kl_unit_gaussian(mean: Tensor, logvar: Tensor) -> Tensor:
    """iodine.py:653-659."""
    return 0.5 * (torch.exp(logvar) + mean ** 2 - 1 - logvar)


def sample(mean: Tensor, logvar: Tensor, eps: Tensor) -> Tensor:
    """iodine.py:620-634 with the noise made explicit."""
    return mean + torch.exp(0.5 * logvar) * eps


def layernorm(x: Tensor) -> Tensor:
    """iodine.py:376-395 (no affine; biased std for 5-D, torch.std for 3-D)."""
    if x.dim() == 3:
        m = x.mean(dim=2, keepdim=True)
        s = x.std(dim=2, keepdim=True)
    elif x.dim() == 5:
        m = x.mean(dim=(2, 3, 4), keepdim=True)
        s = torch.sqrt(((x - m) ** 2).mean(dim=(2, 3, 4), keepdim=True))
    else:
        raise ValueError('invalid size for layernorm')
    return (x - m) / (s + 1e-5)


def elbo_terms(x: Tensor, post_mean: Tensor, post_logvar: Tensor, eps: Tensor,
               p: Dict[str, Tensor], a: Arch) -> Dict[str, Tensor]:
    """One ``IODINE.elbo`` evaluation.  iodine.py:161-241."""
    z = sample(post_mean, post_logvar, eps)
    mean, logits = decoder(z, p, a)
    mask = F.softmax(logits, dim=1)
    kl = kl_unit_gaussian(post_mean, post_logvar).mean(0).sum()
    k_ll = gaussian_log_likelihood(x[:, None], mean, a.sigma)            # (B,K,3,S,S)
    ll_px = torch.logsumexp(torch.log(mask + 1e-12) + k_ll, dim=1)       # (B,3,S,S)
    ll = ll_px.mean(0).sum()
    return dict(z=z, mean=mean, logits=logits, mask=mask, k_ll=k_ll, ll_px=ll_px,
                kl=kl, ll=ll, elbo=ll - kl)


def input_encoding(x: Tensor, t: Dict[str, Tensor], post_mean: Tensor, post_logvar: Tensor,
                   g_mean: Tensor, g_mask: Tensor, g_pm: Tensor, g_plv: Tensor,
                   a: Arch) -> Tuple[Tensor, Tensor]:
    """``get_input_encoding``: (B,K,17,S,S), (B,K,4L) for the full encoding list; an entry missing from ``a.encoding`` drops its
    channels (iodine.py:253-340 tests ``in self.encodings`` entry by entry).  Channel order is fixed by the CODE order of
    iodine.py:277-340, not by the order of the list."""
    B, K = post_mean.shape[:2]
    S = a.img_size
    ln = layernorm if a.layernorm else (lambda v: v)
    lat = []
    if 'posterior' in a.encoding:
        lat += [post_mean, post_logvar]
    if 'grad_post' in a.encoding:
        lat += [ln(g_pm), ln(g_plv)]
    latent = torch.cat(lat, dim=-1) if lat else post_mean.new_zeros(B, K, 0)
    k_like = torch.exp(t['k_ll'].sum(dim=2, keepdim=True))              # (B,K,1,S,S)
    mask_post = k_like / k_like.sum(dim=1, keepdim=True)
    like = torch.exp(t['ll_px'].sum(dim=1, keepdim=True))[:, None].expand(B, K, 1, S, S)
    mixture = (t['mask'] * k_like).sum(dim=1, keepdim=True)
    loo = (mixture - t['mask'] * k_like) / (1 - t['mask'] + 1e-5)
    coords = coord_planes(S, x.dtype, x.device)[None, None].expand(B, K, 2, S, S)
    parts = (
        ('image', x[:, None].expand(B, K, a.img_channels, S, S)),       # 0-2  image
        ('means', t['mean']),                                            # 3-5  means
        ('mask', t['mask']),                                             # 6    mask
        ('mask_logits', t['logits']),                                    # 7    mask_logits
        ('mask_posterior', mask_post),                                   # 8    mask_posterior
        ('grad_means', ln(g_mean)),                                      # 9-11 grad_means
        ('grad_mask', ln(g_mask)),                                       # 12   grad_mask
        ('likelihood', ln(like.contiguous())),                           # 13   likelihood
        ('leave_one_out_likelihood', ln(loo)),                           # 14   leave_one_out_likelihood
        ('coordinate', coords),                                          # 15-16 coordinate
    )
    enc = torch.cat([v for k, v in parts if k in a.encoding], dim=2)
    return enc.detach(), latent.detach()


def refine(enc: Tensor, latent: Tensor, hidden: Optional[Tuple[Tensor, Tensor]],
           p: Dict[str, Tensor], a: Arch, keep: Optional[List[Tensor]] = None):
    """``RefinementNetwork.forward``.  iodine.py:466-503.
    Returns (delta_mean, delta_logvar, (h1, c1))."""
    B, K = enc.shape[:2]
    v = enc.reshape(B * K, *enc.shape[2:])
    lat = latent.reshape(B * K, -1)
    v = conv_stack(v, p, 'refine.mlc', a.ref_layers, a.ref_kernel, a.ref_stride, keep)
    v = F.adaptive_avg_pool2d(v, (1, 1)).reshape(B * K, -1)
    v = F.elu(F.elu(F.linear(v, p['refine.mlp.layers.0.weight'], p['refine.mlp.layers.0.bias'])))
    v = torch.cat((v, lat), dim=1)
    H = a.ref_mlp
    if hidden is None:
        h0 = torch.zeros(B * K, H, dtype=v.dtype)
        c0 = torch.zeros(B * K, H, dtype=v.dtype)
    else:
        h0, c0 = hidden
    gates = (F.linear(v, p['refine.lstm.weight_ih'], p['refine.lstm.bias_ih'])
             + F.linear(h0, p['refine.lstm.weight_hh'], p['refine.lstm.bias_hh']))
    gi, gf, gg, go = gates.chunk(4, dim=1)                     # torch gate order i, f, g, o
    c1 = torch.sigmoid(gf) * c0 + torch.sigmoid(gi) * torch.tanh(gg)
    h1 = torch.sigmoid(go) * torch.tanh(c1)
    d_mean = F.linear(c1, p['refine.mean_update.weight'], p['refine.mean_update.bias'])
    d_logvar = F.linear(c1, p['refine.logvar_update.weight'], p['refine.logvar_update.bias'])
    L = a.dim_latent
    return d_mean.reshape(B, K, L), d_logvar.reshape(B, K, L), (h1, c1)


# --------------------------------------------------------------------------
# the refinement loop
# --------------------------------------------------------------------------

def _loop(x: Tensor, eps: Tensor, p: Dict[str, Tensor], a: Arch, training: bool,
          trace: Optional[List[Dict[str, Tensor]]] = None):
    """Shared body of ``IODINE.forward`` (iodine.py:115-158, training=True) and
    ``IODINE.encode`` (iodine.py:73-105, training=False)."""
    B = x.shape[0]
    K, L, T = a.slots, a.dim_latent, a.iters
    pm = p['posterior.init_mean'][None, None].repeat(B, K, 1)       # iodine.py:607-618
    plv = p['posterior.init_logvar'][None, None].repeat(B, K, 1)
    if not pm.requires_grad:
        pm.requires_grad_(True)
        plv.requires_grad_(True)
    hidden = None
    elbos, kls, lls = [], [], []
    for i in range(T):
        t = elbo_terms(x, pm, plv, eps[i], p, a)
        # (B * elbo).backward(): only these four gradients are consumed (iodine.py:90,137,
        # 181,187,265-267,296,301); parameter gradients of the inner backward are wiped by
        # zero_grad (lib/engine/train.py:62).
        g_mean, g_mask, g_pm, g_plv = torch.autograd.grad(
            B * t['elbo'], [t['mean'], t['mask'], pm, plv], retain_graph=training)
        elbos.append(t['elbo']); kls.append(t['kl']); lls.append(t['ll'])
        enc, latent = input_encoding(x, t, pm, plv, g_mean, g_mask, g_pm, g_plv, a)
        d_mean, d_logvar, hidden = refine(enc, latent, hidden, p, a)
        if trace is not None:
            trace.append(dict(
                z=t['z'].detach(), mean=t['mean'].detach(), logits=t['logits'].detach(),
                mask=t['mask'].detach(), elbo=t['elbo'].detach(), kl=t['kl'].detach(),
                ll=t['ll'].detach(), g_mean=g_mean, g_mask=g_mask, g_pm=g_pm, g_plv=g_plv,
                enc=enc, latent=latent, d_mean=d_mean.detach(), d_logvar=d_logvar.detach(),
                h1=hidden[0].detach(), c1=hidden[1].detach(),
                post_mean=pm.detach(), post_logvar=plv.detach()))
        if not training:
            d_mean, d_logvar = d_mean.detach(), d_logvar.detach()       # iodine.py:99
            hidden = (hidden[0].detach(), hidden[1].detach())           # graph freed (retain_graph=False)
        pm = pm.detach() + d_mean                                       # iodine.py:642-643
        plv = plv.detach() + d_logvar
        if not pm.requires_grad:
            pm.requires_grad_(True)
            plv.requires_grad_(True)
    return pm, plv, elbos, kls, lls


def train_forward(x: Tensor, eps: Tensor, p: Dict[str, Tensor], a: Arch,
                  trace: Optional[List[Dict[str, Tensor]]] = None) -> Dict[str, Tensor]:
    """``loss = model(x)``: -sum_i (i+1)/(T+1) ELBO_i over T+1 evaluations.  iodine.py:115-158.
    ``eps`` has shape (T+1, B, K, L)."""
    pm, plv, elbos, kls, lls = _loop(x, eps, p, a, True, trace)
    t = elbo_terms(x, pm, plv, eps[a.iters], p, a)
    elbos.append(t['elbo']); kls.append(t['kl']); lls.append(t['ll'])
    n = len(elbos)
    total = 0
    for i, e in enumerate(elbos):
        total = total + (i + 1) / n * e
    return dict(loss=-total, elbos=torch.stack(elbos), kls=torch.stack(kls),
                lls=torch.stack(lls), post_mean=pm, post_logvar=plv,
                final_mask=t['mask'], final_mean=t['mean'])


def train_step_grads(x: Tensor, eps: Tensor, p: Dict[str, Tensor], a: Arch):
    """lib/engine/train.py:60-63 without the optimizer: loss and d loss / d params."""
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    out = train_forward(x, eps, q, a)
    names = list(q.keys())
    grads = torch.autograd.grad(out['loss'], [q[n] for n in names], allow_unused=True)
    gd = {n: (g if g is not None else torch.zeros_like(q[n])) for n, g in zip(names, grads)}
    det = {k: (v.detach() if isinstance(v, Tensor) else v) for k, v in out.items()}
    return det, gd


def reconstruct(x: Tensor, eps: Tensor, p: Dict[str, Tensor], a: Arch,
                trace: Optional[List[Dict[str, Tensor]]] = None) -> Dict[str, Tensor]:
    """``IODINE.reconstruct``: T refinement iterations, one more sample, decode.
    iodine.py:59-112.  ``eps`` has shape (T+1, B, K, L): T inside ``elbo`` plus the final draw."""
    q = {k: v.detach() for k, v in p.items()}
    pm, plv, elbos, kls, lls = _loop(x, eps, q, a, False, trace)
    with torch.no_grad():
        z = sample(pm, plv, eps[a.iters])
        mean, logits = decoder(z, q, a)
        mask = F.softmax(logits, dim=1)
        pred = torch.sum(mask * mean, dim=1)
    return dict(pred=pred, mask=mask, mean=mean, z=z, post_mean=pm.detach(),
                post_logvar=plv.detach(), elbos=torch.stack([e.detach() for e in elbos]),
                kls=torch.stack([k.detach() for k in kls]),
                lls=torch.stack([l.detach() for l in lls]))


# --------------------------------------------------------------------------
# closed forms of the inner backward (used to cross-check the autograd path and
# as per-stage references for the HIP pixel kernel)
# --------------------------------------------------------------------------

def pixel_closed_form(x: Tensor, mean: Tensor, logits: Tensor, sigma: float) -> Dict[str, Tensor]:
    """d(B*ELBO)/d mean, d/d mask and d/d (pre-sigmoid rgb, logits) in closed form
    (SURVEY.md rows G1, G2).  Follows the same definitions as iodine.py:185-216."""
    mask = F.softmax(logits, dim=1)
    k_ll = gaussian_log_likelihood(x[:, None], mean, sigma)
    a = torch.log(mask + 1e-12) + k_ll
    ll_px = torch.logsumexp(a, dim=1, keepdim=True)
    r = torch.exp(a - ll_px)                                   # responsibilities (B,K,3,S,S)
    g_mean = r * (x[:, None] - mean) / (sigma ** 2)
    g_mask = (r / (mask + 1e-12)).sum(dim=2, keepdim=True)
    d_rgb = g_mean * mean * (1 - mean)
    d_logit = mask * (g_mask - (mask * g_mask).sum(dim=1, keepdim=True))
    return dict(mask=mask, g_mean=g_mean, g_mask=g_mask, d_rgb=d_rgb, d_logit=d_logit,
                ll_px=ll_px[:, 0])
