"""The module boundary beyond forward / reconstruct: the public ``elbo(x)``, the logger side
channel and the state ``elbo()`` leaves on ``self`` against values captured from the reference
(tests/golden/tiny_logger.npz), call-order errors of the saved forward, the library's own normal generator against its
numpy restatement, hipGraph replay, every non-default kernel option end to end, the reference-written checkpoint, and the
0/0 edge of the un-stabilised mask posterior (iodine.py:286-293)."""
import dataclasses
import os

import numpy as np
import pytest
import torch

from iodine_amd import _lib, checkpoint, synth
from iodine_amd.model import logger
from oracle import iodine_oracle as O
from oracle import philox_oracle as P
from util import (GOLDEN, check_trajectory_params, golden_setup, load_golden, make_hip_model, rel_err, rel_l2,
                  trajectory_setup)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _check_logger(lg, tag, K, with_init):
    keys = ['image', 'pred', 'kl', 'likelihood'] + [f'mask_{i}' for i in range(K)] + [f'pred_{i}' for i in range(K)]
    if with_init:
        keys += ['init_mean', 'init_logvar']
    for k in keys:
        assert k in logger, k
        ref = lg[f'{tag}.logger.{k}']
        got = logger[k].detach().double().cpu().numpy()
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        scale = max(float(np.abs(ref).max()), 1e-3)
        assert float(np.abs(got - ref).max()) <= 2e-4 * scale, (tag, k, float(np.abs(got - ref).max()), scale)


def _check_self(m, lg, tag):
    for k in ('z', 'mean', 'mask', 'mask_logits'):
        assert rel_err(getattr(m, k).cpu(), lg[f'{tag}.self.{k}']) < 2e-4, (tag, k)


def test_logger_side_channel_and_module_state_match_reference():
    g, lg = load_golden('tiny'), load_golden('tiny_logger')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    logger.things.clear()
    m.train()
    m(x.to(DEV), eps.to(DEV))                                             # iodine.py:156-157 + the final elbo(), :226-239
    _check_logger(lg, 'train', arch.slots, with_init=True)
    _check_self(m, lg, 'train')
    logger.things.clear()
    m.eval()
    m.reconstruct(x.to(DEV), eps.to(DEV))                                 # the LAST elbo() of encode, not the final decode
    _check_logger(lg, 'recon', arch.slots, with_init=False)
    _check_self(m, lg, 'recon')


def test_public_elbo_matches_reference_and_oracle():
    g, lg = load_golden('tiny'), load_golden('tiny_logger')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    logger.things.clear()
    v = m.elbo(x.to(DEV), eps[0].to(DEV))                                 # fresh module: initial posterior (init_unit)
    ref = float(lg['elbo.value'])
    assert abs(v.item() - ref) <= 1e-5 * abs(ref)
    assert abs(v.item() - float(g['f32.train.elbos'][0])) <= 1e-5 * abs(ref)
    _check_logger(lg, 'elbo', arch.slots, with_init=False)
    _check_self(m, lg, 'elbo')
    # after a reconstruct the posterior is lambda_T: elbo(x) then evaluates THAT posterior (like the reference's self.posterior)
    m.reconstruct(x.to(DEV), eps.to(DEV))
    pm, plv = m.posterior.mean.cpu(), m.posterior.logvar.cpu()
    v2 = m.elbo(x.to(DEV), eps[1].to(DEV))
    t = O.elbo_terms(x, pm, plv, eps[1], params, arch)
    assert abs(v2.item() - t['elbo'].item()) <= 1e-4 * abs(t['elbo'].item())
    assert rel_err(m.elbo_terms[0, 1:].cpu(), torch.stack([t['kl'], t['ll']]).detach()) < 1e-4
    v3 = m.elbo(x.to(DEV))                                                # eps=None: library generator, finite and different
    assert np.isfinite(v3.item()) and v3.item() != v2.item()


def test_posterior_after_training_forward_is_lambda_T():
    """Gaussian.update (iodine.py:636-645) leaves lambda_T on the reference's module after ``model(x)``; a following
    ``model.elbo(x)`` samples from it (iodine.py:170) - in one call and through the chunked path."""
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    ref = O.train_forward(x, eps, params, arch)
    for cap in (0, 1):
        m = make_hip_model(arch, params)
        if cap:
            m.set_option('batch_cap', cap)
        assert m.posterior.mean is None
        m(x.to(DEV), eps.to(DEV))
        assert rel_err(m.posterior.mean.cpu(), ref['post_mean'].detach()) < 1e-4
        assert rel_err(m.posterior.logvar.cpu(), ref['post_logvar'].detach()) < 1e-4
        v = m.elbo(x.to(DEV), eps[1].to(DEV))
        t = O.elbo_terms(x, ref['post_mean'].detach(), ref['post_logvar'].detach(), eps[1], params, arch)
        assert abs(v.item() - t['elbo'].item()) <= 1e-4 * abs(t['elbo'].item())


def test_graph_mode_replays_with_fresh_caller_tensors():
    """Option graph=1: the library only sees the module's persistent staging buffers, so steps whose inputs / outputs are new
    allocations every time (a data loader's batches, autograd's gradient buffers) still replay ONE captured graph per entry
    point instead of re-capturing; results are bitwise those of the eager path, and what the caller got back is never
    overwritten by the next step."""
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    m0 = make_hip_model(arch, params)
    m0.zero_grad(set_to_none=True)
    l0 = m0(x.to(DEV), eps.to(DEV)); l0.backward()
    g0 = [p.grad.clone() for p in m0.parameters()]
    r0 = m0.reconstruct(x.to(DEV), eps.to(DEV))
    m = make_hip_model(arch, params)
    m.set_option('graph', 1)
    keep = []
    with torch.cuda.stream(torch.cuda.Stream()):
        for step in range(5):
            xd, ed = x.clone().to(DEV), eps.clone().to(DEV)               # fresh device allocations every step
            junk = torch.empty(1000 + 37 * step, device=DEV)              # shifts what the caching allocator hands out next
            m.zero_grad(set_to_none=True)
            loss = m(xd, ed)
            loss.backward()
            keep.append((loss.detach(), [p.grad for p in m.parameters()], m.reconstruct(xd, ed), junk))
        torch.cuda.synchronize()
    for loss, grads, rec, _ in keep:
        assert torch.equal(loss, l0.detach())
        assert all(torch.equal(a, b) for a, b in zip(grads, g0))
        assert all(torch.equal(a, b) for a, b in zip(rec, r0))
    # three entry points (train forward, train backward, reconstruct): one capture each (second call), then replays
    assert m.profile_read('graph_captures')[1] == 3 and m.profile_read('graph_replays')[1] == 9


def test_saved_forward_has_an_identity():
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    xd, ed = x.to(DEV), eps.to(DEV)
    loss = m(xd, ed)
    m.reconstruct(xd, ed)                                                 # re-uses the workspace
    with pytest.raises(RuntimeError, match='stale forward'):
        loss.backward()
    l1 = m(xd, ed)
    l2 = m(xd, ed)
    with pytest.raises(RuntimeError, match='stale forward'):
        l1.backward()                                                     # the reference would hold two graphs
    m.zero_grad(set_to_none=True)
    l2.backward()
    g2 = [p.grad.clone() for p in m.parameters()]
    with pytest.raises(RuntimeError):
        l2.backward()                                                     # no retain_graph: the saved forward is consumed
    m.zero_grad(set_to_none=True)
    m(xd, ed).backward()
    assert all(torch.equal(a, p.grad) for a, p in zip(g2, m.parameters()))
    # the raw C ABI reports the same conditions as status codes
    L = _lib.lib()
    flat = torch.zeros(sum(p.numel() for p in m.parameters()), device=DEV)
    rc = L.iodine_train_backward_flat(m._handle, None, None, _lib.ptr(flat), 0)
    assert rc == 3 and b'no iodine_train_forward' in L.iodine_last_error(m._handle)


def test_params_dirty_hook():
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    xd, ed = x.to(DEV), eps.to(DEV)
    e0 = m.elbo(xd, ed[0]).item()
    with torch.no_grad():
        for p in m.decoder.parameters():
            p.data.mul_(0.5)                                              # bypasses the version counter
    assert m.elbo(xd, ed[0]).item() == e0                                 # documented: not seen ...
    m.mark_params_dirty()
    assert m.elbo(xd, ed[0]).item() != e0                                 # ... until the caller says so


def test_library_normals_match_numpy_restatement():
    L = _lib.lib()
    for n, seed, stream in ((4096, 1234, 0), (1001, 2 ** 40 + 17, 5), (3, 0, 2 ** 33 + 1)):
        out = torch.empty(n, device=DEV)
        _lib.check(L.iodine_randn(None, _lib.ptr(out), n, seed, stream), None, 'iodine_randn')
        ref = P.randn(n, seed, stream)
        assert np.abs(out.cpu().numpy() - ref).max() < 2e-5, (n, seed, stream)
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params).manual_seed(99)
    a = m.encode(x.to(DEV))
    b = m.encode(x.to(DEV))
    assert not torch.equal(a, b)                                          # a new draw per call
    m.manual_seed(99)
    assert torch.equal(m.encode(x.to(DEV)), a)                            # reproducible from the seed
    e = torch.from_numpy(P.randn(eps.numel(), 99, 0)).view_as(eps)        # the first draw, through the oracle's generator
    assert rel_err(a.cpu(), O.reconstruct(x, e, params, arch)['z']) < 1e-3


@pytest.mark.parametrize('case', ['tiny', 'cfg2_dsprites_k6_t5_b2'])
def test_hip_graph_replay_is_bit_identical(case):
    g = load_golden(case)
    arch, params, x, eps, _ = golden_setup(g)
    xd, ed = x.to(DEV), eps.to(DEV)
    m = make_hip_model(arch, params)
    ref_out = [t.clone() for t in m.reconstruct(xd, ed)]
    m.zero_grad(set_to_none=True)
    ref_loss = m(xd, ed)
    ref_loss.backward()
    ref_grads = [p.grad.clone() for p in m.parameters()]
    m.set_option('graph', 1)
    for rep in range(4):                                                  # eager, capture, replay, replay
        out = m.reconstruct(xd, ed)
        assert all(torch.equal(a, b) for a, b in zip(out, ref_out)), rep
        m.zero_grad(set_to_none=True)
        loss = m(xd, ed)
        loss.backward()
        assert torch.equal(loss, ref_loss), rep
        assert all(torch.equal(p.grad, r) for p, r in zip(m.parameters(), ref_grads)), rep
    m.set_option('graph', 0)


OPTIONS = [('conv_precision', 0), ('conv_variant', 1), ('out_bwd_fused', 0), ('fuse_l0', 0), ('refine_split', 0), ('head_fused', 0), ('refine_bwd_fused', 0), ('refine_ws', 0), ('refine_l0_fused', 0), ('head_mfma', 0), ('wgrad_accum', 1), ('dec_out_rows', 0), ('graph', 1)]


@pytest.mark.parametrize('case', ['cfg1_dsprites_k4_t3_b4', 'cfg3_clevr_k7_t5_b1'])
@pytest.mark.parametrize('opt,val', OPTIONS)
def test_every_kernel_option_end_to_end(opt, val, case):
    """Each non-default kernel selection through a whole reconstruct + training step against the reference goldens: cfg1
    (64 x 64, 32 channels) and the cfg3 golden (128 x 128, 64 channels, B = 1 - where the weight-stationary / LDS-tiled / exact
    kernels differ most)."""
    g = load_golden(case)
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    m.set_option(opt, val)
    xd, ed = x.to(DEV), eps.to(DEV)
    pred, mask, mean = m.reconstruct(xd, ed)
    assert rel_err(m.elbo_terms[:, 0].cpu(), g['f32.recon.elbos']) < 1e-4
    a = pred.double().cpu().flatten()
    assert abs(float((a * a).sum()) - float(g['f32.recon.pred.sumsq'])) <= 1e-4 * float(g['f32.recon.pred.sumsq'])
    m.zero_grad(set_to_none=True)
    loss = m(xd, ed)
    loss.backward()
    assert abs(loss.item() - float(g['f64.train.loss'])) <= 1e-4 * abs(float(g['f64.train.loss']))
    bad = []
    for n, p in m.named_parameters():
        a = p.grad.double().cpu().flatten()
        ss, ref_ss = float((a * a).sum()), float(g[f'f64.train.grad.{n}.sumsq'])
        if abs(ss - ref_ss) > 2e-3 * ref_ss + 1e-12:
            bad.append((n, ss, ref_ss))
    assert not bad, bad


@pytest.mark.parametrize('case', ['cfg1_dsprites_k4_t3_b4', 'cfg3_clevr_k7_t5_b1'])
def test_wgrad_accum_on_the_exact_fp32_path(case):
    """Option wgrad_accum has its own read-modify-write tail in conv3x3_wgrad_f32_ws_kernel, and with conv_precision 0 the output conv
    keeps its per-pass reduction while layers 1.. accumulate over the passes: the mix against the reference goldens, at
    one 32-channel and one 64-channel shape."""
    g = load_golden(case)
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params, options={'conv_precision': 0, 'wgrad_accum': 1})
    ref = make_hip_model(arch, params, options={'conv_precision': 0})
    xd, ed = x.to(DEV), eps.to(DEV)
    for mm in (m, ref):
        mm.zero_grad(set_to_none=True)
        mm(xd, ed).backward()
    bad = []
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        a = p.grad.double().cpu().flatten()
        ss, ref_ss = float((a * a).sum()), float(g[f'f64.train.grad.{n}.sumsq'])
        if abs(ss - ref_ss) > 2e-3 * ref_ss + 1e-12:
            bad.append((n, ss, ref_ss))
        assert rel_l2(p.grad.cpu(), q.grad.cpu()) < 2e-5, n          # same products, other summation order over the passes
    assert not bad, bad


@pytest.mark.parametrize('case', ['cfg1_dsprites_k4_t3_b4', 'cfg3_clevr_k7_t5_b1'])
def test_fused_output_conv_backward_equals_the_two_kernel_form(case):
    """Training: dec_out_bwd_fused_f16x3_kernel (one pass over the saved activation) against dec_out_dgrad_f16x3_kernel +
    dec_out_wgrad_gemm_f16x3_kernel (same packs, same three split passes; only tile shapes / summation order differ)."""
    g = load_golden(case)
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    xd, ed = x.to(DEV), eps.to(DEV)
    grads = []
    for fused in (1, 0):
        m.set_option('out_bwd_fused', fused)
        m.zero_grad(set_to_none=True)
        loss = m(xd, ed)
        loss.backward()
        grads.append((loss.item(), {n: p.grad.double().cpu() for n, p in m.named_parameters()}))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * abs(grads[1][0])
    for n in grads[0][1]:
        a, b = grads[0][1][n], grads[1][1][n]
        assert float((a - b).norm()) <= 2e-5 * float(b.norm()) + 1e-12, (n, float((a - b).norm()), float(b.norm()))


def test_reference_written_checkpoint_continues_on_the_gpu():
    """tests/golden/ckpt_tiny was written by the reference's Checkpointer after two steps of traj_tiny: loading it and taking
    steps 3 and 4 with the HIP step + fused Adam lands on the reference's losses and final parameters."""
    from iodine_amd.optim import make_optimizer
    tr, arch, params, x, eps = trajectory_setup('traj_tiny')
    m = make_hip_model(arch, {k: torch.zeros_like(v) for k, v in params.items()})
    opt = make_optimizer(m, base_lr=float(tr['meta_lr']))
    extra = checkpoint.load_checkpoint(checkpoint.last_checkpoint(os.path.join(GOLDEN, 'ckpt_tiny')), m, opt)
    assert extra == {'epoch': 2}
    xd = x.to(DEV)
    for s in (2, 3):
        loss = m(xd, eps[s].to(DEV))
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert abs(loss.item() - float(tr['f64.losses'][s])) <= 2e-5 * abs(float(tr['f64.losses'][s])), s
    check_trajectory_params(tr, 'f32', m.named_parameters(), 0.05)
    assert all(int(opt.state[p]['step']) == 4 for p in m.parameters())


def test_mask_posterior_zero_over_zero_edge():
    """iodine.py:286-293 divides exp(sum_c loglik) by its sum over slots WITHOUT max-subtraction: when every slot's summed
    log-likelihood underflows (x = 1 where every slot predicts < 0.03 at sigma 0.1: sum_c loglik < -142, far below the
    log of the smallest denormal) the channel is 0/0 = NaN and poisons that image's refinement.  The HIP path and the oracle must agree on WHERE non-finite
    values appear and on every finite value; the other image of the batch must not be touched."""
    arch = O.tiny_arch(slots=3, iters=2)
    pn = synth.make_params(O.param_shapes(arch), seed=5, dec_gain=2.0, posterior_scale=0.05)
    pn['decoder.conv.bias'] = np.array([-6.0, -6.0, -6.0, 0.0], dtype=np.float32)        # rgb: 3e-4 .. 0.03 in every slot
    params = {k: torch.from_numpy(v) for k, v in pn.items()}
    x = torch.from_numpy(synth.make_images(2, arch.img_size, seed=3, kind='blobs')[0]) * 0.3   # image 1: dim scene, sum_c loglik > -7
    x[0] = 1.0                                                                           # image 0: far from every slot
    eps = torch.from_numpy(synth.make_eps(arch.iters, 2, arch.slots, arch.dim_latent, seed=4))
    trace = []
    ref = O.reconstruct(x, eps, params, arch, trace=trace)
    m = make_hip_model(arch, params)
    m.set_option('stop_after_iters', 1)
    m.reconstruct(x.to(DEV), eps.to(DEV))
    m.set_option('stop_after_iters', -1)
    B, K, S = 2, arch.slots, arch.img_size
    enc = m.debug_buffer('enc').cpu().view(B, K, S, S, 20)[..., :17].permute(0, 1, 4, 2, 3)
    ref_enc = trace[0]['enc']
    assert torch.isnan(ref_enc[0, :, 8]).all() and torch.isfinite(ref_enc[1]).all()     # the edge is really hit
    assert torch.equal(torch.isfinite(enc), torch.isfinite(ref_enc))
    assert torch.equal(torch.isnan(enc), torch.isnan(ref_enc))
    fin = torch.isfinite(ref_enc)
    for c in range(17):
        for b in range(B):
            f = fin[b, :, c]
            if f.any():
                assert rel_err(enc[b, :, c][f], ref_enc[b, :, c][f]) < 5e-4, (b, c)
    pred, mask, mean = m.reconstruct(x.to(DEV), eps.to(DEV))
    for got, want in ((pred.cpu(), ref['pred']), (mask.cpu(), ref['mask']), (mean.cpu(), ref['mean']),
                      (m.posterior.mean.cpu(), ref['post_mean'])):
        assert torch.equal(torch.isfinite(got), torch.isfinite(want))
        assert not torch.isfinite(want[0]).any() and torch.isfinite(want[1]).all()
        assert rel_err(got[1], want[1]) < 2e-4
    e, er = m.elbo_terms[:, 0].cpu(), ref['elbos']
    assert torch.equal(torch.isfinite(e), torch.isfinite(er)) and torch.isfinite(er[0]) and not torch.isfinite(er[1])
    assert abs(e[0].item() - er[0].item()) <= 1e-4 * abs(er[0].item())
    loss = m(x.to(DEV), eps.to(DEV))
    assert not torch.isfinite(loss).item() and not torch.isfinite(O.train_forward(x, eps, params, arch)['loss']).item()


@pytest.mark.parametrize('B,K,T', [(3, 6, 2), (5, 6, 2), (2, 11, 2)])
def test_odd_batches_and_eleven_slots_at_64px(B, K, T):
    """dSprites architecture (64x64) with batch sizes that are not a multiple of anything and with K = 11."""
    arch = O.dsprites_arch(slots=K, iters=T)
    pn = synth.make_params(O.param_shapes(arch), seed=31 + B, dec_gain=3.0, posterior_scale=0.05)
    params = {k: torch.from_numpy(v) for k, v in pn.items()}
    imgs, _ = synth.make_images(B, arch.img_size, seed=B, kind='blobs')
    x = torch.from_numpy(imgs)
    eps = torch.from_numpy(synth.make_eps(T, B, K, arch.dim_latent, seed=K))
    ref = O.reconstruct(x, eps, params, arch)
    m = make_hip_model(arch, params)
    pred, mask, mean = m.reconstruct(x.to(DEV), eps.to(DEV))
    assert rel_err(m.elbo_terms[:, 0].cpu(), ref['elbos']) < 1e-4
    assert rel_err(pred.cpu(), ref['pred']) < 2e-4 and rel_err(mask.cpu(), ref['mask']) < 2e-4
    out, grads = O.train_step_grads(x, eps, params, arch)
    m.zero_grad(set_to_none=True)
    loss = m(x.to(DEV), eps.to(DEV))
    loss.backward()
    assert abs(loss.item() - out['loss'].item()) <= 1e-5 * abs(out['loss'].item())
    bad = [(n, rel_l2(p.grad.cpu().numpy(), grads[n].numpy())) for n, p in m.named_parameters()
           if not rel_l2(p.grad.cpu().numpy(), grads[n].numpy()) < 1e-3]
    assert not bad, bad


def test_batches_beyond_one_library_call_run_in_chunks():
    """IODINE.max_batch: the kernels use 32-bit element offsets, so one library call takes a bounded number of images (145 at
    cfg3); the wrapper runs larger batches as chunks of independent images.  With ``batch_cap`` = 2 a batch of 5 goes through
    the chunked paths of reconstruct / elbo / decode / forward + backward and must agree with the single call: per-image
    outputs bit for bit, batch means and gradients to summation order."""
    g = load_golden('cfg1_dsprites_k4_t3_b4')
    arch, params, _, _, _ = golden_setup(g)
    B, K, T, L = 5, arch.slots, arch.iters, arch.dim_latent
    imgs, _ = synth.make_images(B, arch.img_size, seed=77, kind='blobs')
    x = torch.from_numpy(imgs).to(DEV)
    eps = torch.from_numpy(synth.make_eps(T, B, K, L, seed=78)).to(DEV)
    m = make_hip_model(arch, params)
    assert m.max_batch() > 1000 and m.max_batch(training=True) <= m.max_batch()

    def run_all():
        out = {}
        out['recon'] = m.reconstruct(x, eps)
        out['state'] = (m.z, m.mean, m.mask, m.mask_logits, m.posterior.mean, m.posterior.logvar)
        out['recon_terms'] = m.elbo_terms.clone()
        out['log_recon'] = {k: logger[k].clone() for k in ('image', 'pred', 'mask_0', f'pred_{K - 1}')}
        out['elbo'] = m.elbo(x, eps[0]).clone()                       # from the posterior reconstruct left behind
        out['decode'] = m.decode(out['state'][0])
        m.zero_grad(set_to_none=True)
        loss = m(x, eps)
        out['train_terms'] = m.elbo_terms.clone()
        out['train_z'] = m.z
        (2.0 * loss).backward()
        out['loss'] = loss.detach().clone()
        out['grads'] = {n: p.grad.clone() for n, p in m.named_parameters()}
        out['log_train'] = {k: logger[k].clone() for k in ('image', 'pred', 'kl', 'likelihood', 'init_mean', 'init_logvar')}
        return out

    whole = run_all()
    m.set_option('batch_cap', 2)
    assert m.max_batch() == 2 and IODINE_chunks(m, B) == [(0, 2), (2, 4), (4, 5)]
    parts = run_all()
    m.set_option('batch_cap', 0)
    for a, b in zip(whole['recon'] + whole['state'] + whole['decode'], parts['recon'] + parts['state'] + parts['decode']):
        assert torch.equal(a, b)
    assert torch.equal(whole['train_z'], parts['train_z'])
    for k in whole['log_recon']:
        assert torch.equal(whole['log_recon'][k], parts['log_recon'][k]), k
    for k in ('image', 'pred', 'init_mean', 'init_logvar'):
        assert torch.equal(whole['log_train'][k], parts['log_train'][k]), k
    for k in ('kl', 'likelihood'):
        assert rel_err(parts['log_train'][k].cpu(), whole['log_train'][k].cpu()) < 1e-5, k
    for k in ('recon_terms', 'train_terms', 'elbo', 'loss'):
        assert rel_err(parts[k].cpu(), whole[k].cpu()) < 1e-5, k
    for n in whole['grads']:
        a, b = parts['grads'][n].double().cpu(), whole['grads'][n].double().cpu()
        assert float((a - b).norm()) <= 2e-5 * float(b.norm()) + 1e-12, (n, float((a - b).norm()), float(b.norm()))


def IODINE_chunks(m, B):
    return m._chunks(B, m.max_batch())


def test_profile_stride_brackets_every_nth_dominant_launch():
    """option profile_stride (bench.py: 7): level-1 profiling brackets every n-th launch of a dominant category with HIP events - the
    bracketed count and the 'seen:' count of iodine_profile_read must be consistent, and the result of the step must not depend on it."""
    g = load_golden('cfg3_clevr_k7_t5_b1')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    xd, ed = x.to(DEV), eps.to(DEV)
    ref = m.reconstruct(xd, ed)[0].clone()
    m.profile_read('conv_tile_fwd'); m.profile_read('seen:conv_tile_fwd')
    m.set_option('profile_stride', 4)
    m.set_option('profile', 1)
    out = m.reconstruct(xd, ed)[0]
    torch.cuda.synchronize()
    m.set_option('profile', 0)
    tot, cnt = m.profile_read('conv_tile_fwd')
    _, seen = m.profile_read('seen:conv_tile_fwd')
    layers = arch.dec_layers - 1 if hasattr(arch, 'dec_layers') else 3
    assert seen == (arch.iters + 1) * layers, (seen, cnt)                 # every 64 -> 64 forward launch of T + 1 decodes
    assert cnt == (seen + 3) // 4 and tot > 0.0
    assert torch.equal(out, ref)
    m.set_option('profile_stride', 1)
    with pytest.raises(RuntimeError):
        m.set_option('profile_stride', 0)
