"""Parity at BASELINE.json's full sizes (cfg2: dSprites arch, K=6, T=5, B=32; cfg3: CLEVR6 arch, K=7, T=5, B=32; cfg5
per-GPU shard: K=11, T=7, B=8), where
the CPU oracle takes minutes per step: size-independent properties chain the full batch to cases the reference pinned.

  1. image independence: every image's outputs in the full batch are BITWISE those of a
     batch-of-one run; the batch ELBOs / loss / parameter gradients are the means of the 32 single-image runs;
  2. image 0 of the batch is the reference-generated golden case (same counter-based inputs), so the single-image
     run is compared with lib/modeling/iodine.py's own numbers (tests/golden/cfg3_*.npz, cfg5_*.npz);
  3. one more image of the batch goes through the CPU oracle on this box (seconds at B=1);
  4. determinism: two runs of the full step give bitwise identical losses and gradients;
  5. decode(z) of the latents returned by reconstruct reproduces its outputs bitwise.
"""
import numpy as np
import pytest
import torch

from iodine_amd import synth
from oracle import iodine_oracle as O
from util import golden_setup, load_golden, make_hip_model, rel_err, rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _full_batch(case, B):
    """the golden case's weights / seeds at batch B: image b and eps[:, b] do not depend on B (counter-based RNG)"""
    g = load_golden(case)
    arch, params, x1, eps1, _ = golden_setup(g)
    sw, sx, se = (int(v) for v in g['meta_seeds'])
    kind = str(g['meta_kind'])
    imgs = synth.make_images(B, arch.img_size, seed=sx, kind=kind)
    imgs = imgs[0] if kind == 'blobs' else imgs
    eps = synth.make_eps(arch.iters, B, arch.slots, arch.dim_latent, seed=se)
    x, eps = torch.from_numpy(imgs), torch.from_numpy(eps)
    gB = x1.shape[0]                                                       # batch of the golden case: its images lead the batch
    assert torch.equal(x[:gB], x1) and torch.equal(eps[:, :gB], eps1)
    if case.startswith('cfg2'):
        # uniform-noise images are far from anything the dSprites decoder predicts: a third of them run into the 0/0 of the
        # reference's un-stabilised mask posterior (iodine.py:286-293; the HIP path and the oracle go NaN at the same
        # iteration - tests/test_gpu_boundary.py covers that edge).  The batch-mean identities below need finite images:
        # lower the contrast of the images that are not part of the golden case.
        x[gB:] = 0.5 + 0.35 * (x[gB:] - 0.5)
    return g, arch, params, x, eps


@pytest.mark.parametrize('case,B', [('cfg3_clevr_k7_t5_b1', 32), ('cfg5_clevr_k11_t7_b1', 8), ('cfg2_dsprites_k6_t5_b2', 32)])
def test_full_size_reconstruct(case, B):
    g, arch, params, x, eps = _full_batch(case, B)
    m = make_hip_model(arch, params)
    xd, ed = x.to(DEV), eps.to(DEV)
    pred, mask, mean = m.reconstruct(xd, ed)
    full = m.elbo_terms.clone()
    z = m.encode(xd, ed)                                                   # the final sample (iodine.py:103)
    p2, k2, m2 = m.reconstruct(xd, ed)                                     # determinism
    assert torch.equal(p2, pred) and torch.equal(k2, mask) and torch.equal(m2, mean)
    assert torch.equal(m.elbo_terms, full)
    singles = []
    for b in range(B):
        p, k, mm = m.reconstruct(xd[b:b + 1], ed[:, b:b + 1].contiguous())
        assert torch.equal(p, pred[b:b + 1]) and torch.equal(k, mask[b:b + 1]) and torch.equal(mm, mean[b:b + 1]), b
        singles.append(m.elbo_terms.clone())
    assert rel_err(torch.stack(singles).double().mean(0).cpu(), full.double().cpu()) < 1e-6
    # the leading image(s) == the reference-generated golden case (its ELBOs are the mean over its batch)
    gB = int(g['meta_B'])
    e0 = torch.stack(singles[:gB])[:, :, 0].double().mean(0).cpu().numpy()
    assert (np.abs(e0 - g['f32.recon.elbos']) / np.abs(g['f32.recon.elbos'])).max() < 1e-4
    a = pred[:gB].double().cpu().flatten()
    assert abs(float((a * a).sum()) - float(g['f32.recon.pred.sumsq'])) <= 1e-4 * float(g['f32.recon.pred.sumsq'])
    assert (mask[:gB, :, 0].argmax(dim=1).cpu().numpy() == g['f32.recon.argmax']).mean() >= 0.999
    # another image through the CPU oracle
    b = B - 3
    ref = O.reconstruct(x[b:b + 1], eps[:, b:b + 1], params, arch)
    assert rel_err(singles[b][:, 0].cpu(), ref['elbos']) < 1e-4
    assert rel_err(pred[b:b + 1].cpu(), ref['pred']) < 2e-4
    assert (mask[b, :, 0].argmax(0).cpu() == ref['mask'][0, :, 0].argmax(0)).float().mean() >= 0.999
    if z is not None:
        pd, kd, md = m.decode(z)
        assert torch.equal(pd, pred) and torch.equal(kd, mask) and torch.equal(md, mean)


def _train(m, x, eps):
    m.zero_grad(set_to_none=True)
    loss = m(x, eps)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}


@pytest.mark.parametrize('case,B,prec', [('cfg3_clevr_k7_t5_b1', 32, 1), ('cfg5_clevr_k11_t7_b1', 8, 1), ('cfg2_dsprites_k6_t5_b2', 32, 1),
                                         ('cfg3_clevr_k7_t5_b1', 32, 0), ('cfg2_dsprites_k6_t5_b2', 32, 0)])     # prec 0: the strict exact-fp32 path
def test_full_size_train_step(case, B, prec):
    g, arch, params, x, eps = _full_batch(case, B)
    m = make_hip_model(arch, params, options={'conv_precision': prec})
    xd, ed = x.to(DEV), eps.to(DEV)
    loss, grads = _train(m, xd, ed)
    elbos = m.elbo_terms[:, 0].clone()
    loss2, grads2 = _train(m, xd, ed)                                      # determinism (fixed-order reductions)
    assert torch.equal(loss, loss2) and all(torch.equal(grads[n], grads2[n]) for n in grads)
    # the batch step is the mean of the single-image steps (loss is a batch mean of per-image sums, iodine.py:193,220)
    lsum, gsum, esum = 0.0, {n: torch.zeros_like(v, dtype=torch.float64) for n, v in grads.items()}, 0.0
    gB = int(g['meta_B'])
    first = None
    for b in range(B):
        l1, g1 = _train(m, xd[b:b + 1], ed[:, b:b + 1].contiguous())
        lsum += l1.double()
        esum = esum + m.elbo_terms[:, 0].double()
        for n in gsum:
            gsum[n] += g1[n].double()
        if b == gB - 1:                                                    # mean over the golden case's images so far
            first = ((lsum / gB).clone(), {n: (v / gB).clone() for n, v in gsum.items()})
    assert abs((lsum / B - loss.double()).item()) <= 1e-6 * abs(loss.item())
    assert rel_err((esum / B).cpu(), elbos.double().cpu()) < 1e-6
    bad = [(n, rel_l2(grads[n].cpu().numpy(), (gsum[n] / B).cpu().numpy())) for n in grads
           if not rel_l2(grads[n].cpu().numpy(), (gsum[n] / B).cpu().numpy()) < 2e-5]
    assert not bad, bad
    # the leading image(s) == the reference-generated golden case (fp64 reference gradients)
    l0, g0 = first
    assert abs(l0.item() - float(g['f32.train.loss'])) <= 1e-4 * abs(float(g['f32.train.loss']))
    for n, a in g0.items():
        a = a.double().cpu().flatten()
        ref_ss = float(g[f'f64.train.grad.{n}.sumsq'])
        assert abs(float((a * a).sum()) - ref_ss) <= 2e-3 * ref_ss + 1e-12, n


# ---- BASELINE configs 4 and 5 at their GLOBAL batch on ONE device ---------------------------------------------------------
# cfg4 (CLEVR6, K=7, T=5, B=256 over 8 GPUs) and cfg5 (CLEVR-full shapes, K=11, T=7, B=64 over 8 GPUs) fit one MI355X: cfg4's
# reconstruct is ONE library call just below the 32-bit-offset limit (max_batch 292), its training step crosses the limit
# (max_batch(training) = 187) and runs as two chunks of 128 through _ChunkedTrainStep; cfg5 is one call each (186 / 85).
GLOBAL = [('cfg3_clevr_k7_t5_b1', 256, 32), ('cfg5_clevr_k11_t7_b1', 64, 8)]


@pytest.mark.parametrize('case,B,shard', GLOBAL)
def test_global_batch_reconstruct(case, B, shard):
    g, arch, params, x, eps = _full_batch(case, B)
    m = make_hip_model(arch, params)
    assert m.max_batch() >= B                                              # one library call
    xd, ed = x.to(DEV), eps.to(DEV)
    pred, mask, mean = m.reconstruct(xd, ed)
    full = m.elbo_terms.clone()
    p2, k2, m2 = m.reconstruct(xd, ed)                                     # determinism
    assert torch.equal(p2, pred) and torch.equal(k2, mask) and torch.equal(m2, mean) and torch.equal(m.elbo_terms, full)
    del p2, k2, m2
    # every per-GPU shard of the 8-GPU job, run on its own, is bitwise the same images; the global ELBOs are their mean
    terms = []
    for s in range(0, B, shard):
        p, k, mm = m.reconstruct(xd[s:s + shard], ed[:, s:s + shard].contiguous())
        assert torch.equal(p, pred[s:s + shard]) and torch.equal(k, mask[s:s + shard]) and torch.equal(mm, mean[s:s + shard]), s
        terms.append(m.elbo_terms.double())
    assert rel_err(torch.stack(terms).mean(0).cpu(), full.double().cpu()) < 1e-6
    # image 0 = the reference-generated golden case
    p, k, mm = m.reconstruct(xd[:1], ed[:, :1].contiguous())
    assert torch.equal(p, pred[:1])
    e0 = m.elbo_terms[:, 0].double().cpu().numpy()
    assert (np.abs(e0 - g['f32.recon.elbos']) / np.abs(g['f32.recon.elbos'])).max() < 1e-4
    assert (mask[:1, :, 0].argmax(dim=1).cpu().numpy() == g['f32.recon.argmax']).mean() >= 0.999
    # the LAST image of the batch (highest offsets of the one call) through the CPU oracle
    b = B - 1
    ref = O.reconstruct(x[b:b + 1], eps[:, b:b + 1], params, arch)
    p, k, mm = m.reconstruct(xd[b:b + 1], ed[:, b:b + 1].contiguous())
    assert torch.equal(p, pred[b:b + 1]) and torch.equal(k, mask[b:b + 1])
    assert rel_err(m.elbo_terms[:, 0].cpu(), ref['elbos']) < 1e-4
    assert rel_err(pred[b:b + 1].cpu(), ref['pred']) < 2e-4


@pytest.mark.parametrize('case,B,shard', GLOBAL)
def test_global_batch_train_step(case, B, shard):
    g, arch, params, x, eps = _full_batch(case, B)
    m = make_hip_model(arch, params)
    chunked = B > m.max_batch(training=True)
    assert chunked == (B == 256)                                           # cfg4: two chunks of 128; cfg5: one call
    xd, ed = x.to(DEV), eps.to(DEV)
    loss, grads = _train(m, xd, ed)
    elbos, z, mask = m.elbo_terms.clone(), m.z.clone(), m.mask.clone()
    loss2, grads2 = _train(m, xd, ed)                                      # determinism (fixed-order reductions)
    assert torch.equal(loss, loss2) and all(torch.equal(grads[n], grads2[n]) for n in grads)
    assert torch.equal(m.z, z) and torch.equal(m.mask, mask)
    # the job as the 8 ranks would run it: per-shard steps; the state of the last elbo() is bitwise per image, loss / ELBO terms
    # / gradients are the mean over the shards (what the all-reduce forms)
    lsum, esum = 0.0, 0.0
    gsum = {n: torch.zeros_like(v, dtype=torch.float64) for n, v in grads.items()}
    n_sh = B // shard
    for r in range(n_sh):
        s = r * shard
        l1, g1 = _train(m, xd[s:s + shard], ed[:, s:s + shard].contiguous())
        assert torch.equal(m.z, z[s:s + shard]) and torch.equal(m.mask, mask[s:s + shard]), r
        lsum += l1.double()
        esum = esum + m.elbo_terms.double()
        for n in gsum:
            gsum[n] += g1[n].double()
    assert abs((lsum / n_sh - loss.double()).item()) <= 1e-6 * abs(loss.item())
    assert rel_err((esum / n_sh).cpu(), elbos.double().cpu()) < 1e-6
    bad = [(n, rel_l2(grads[n].cpu().numpy(), (gsum[n] / n_sh).cpu().numpy())) for n in grads
           if not rel_l2(grads[n].cpu().numpy(), (gsum[n] / n_sh).cpu().numpy()) < 2e-5]
    assert not bad, bad
    # image 0 alone == the reference-generated golden case (fp64 reference gradients)
    l0, g0 = _train(m, xd[:1], ed[:, :1].contiguous())
    assert torch.equal(m.z, z[:1])
    assert abs(l0.item() - float(g['f32.train.loss'])) <= 1e-4 * abs(float(g['f32.train.loss']))
    for n, a in g0.items():
        a = a.double().cpu().flatten()
        ref_ss = float(g[f'f64.train.grad.{n}.sumsq'])
        assert abs(float((a * a).sum()) - ref_ss) <= 2e-3 * ref_ss + 1e-12, n
    # the last image of the batch through the CPU oracle (seconds at B = 1): loss, ELBOs and every gradient
    b = B - 1
    out, rg = O.train_step_grads(x[b:b + 1], eps[:, b:b + 1].contiguous(), params, arch)
    l1, g1 = _train(m, xd[b:b + 1], ed[:, b:b + 1].contiguous())
    assert torch.equal(m.z, z[b:b + 1])
    assert abs(l1.item() - float(out['loss'])) <= 1e-4 * abs(float(out['loss']))
    num = sum(float(((g1[n].double().cpu() - rg[n].double()) ** 2).sum()) for n in g1)
    den = sum(float((rg[n].double() ** 2).sum()) for n in g1)
    assert (num / den) ** 0.5 < 1e-3
