"""Parity of the HIP training step (loss = model(x); loss.backward(), lib/engine/train.py:60-63) with the
reference-generated goldens and the CPU oracle: loss, per-iteration ELBOs and every parameter gradient.
Gates: ELBO 1e-3 relative, parameter-gradient rel-L2 1e-3; asserted tighter."""
import numpy as np
import pytest
import torch

from oracle import iodine_oracle as O
from util import golden_setup, load_golden, make_hip_model, rel_err, rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _train_step(m, x, eps):
    m.zero_grad(set_to_none=True)
    loss = m(x.to(DEV), eps.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    return loss


def test_tiny_gradients_full_tensors():
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    loss = _train_step(m, x, eps)
    assert abs(loss.item() - float(g['f32.train.loss'])) <= 1e-5 * abs(float(g['f32.train.loss']))
    assert rel_err(m.elbo_terms[:, 0].cpu(), g['f32.train.elbos']) < 1e-5
    bad = []
    for n, p in m.named_parameters():
        ref = g['f64.train.grad.' + n]
        e = rel_l2(p.grad.cpu().numpy(), ref)
        if not e < 2e-4:
            bad.append((n, e))
    assert not bad, bad


@pytest.mark.parametrize('case', ['cfg1_dsprites_k4_t3_b4', 'cfg2_dsprites_k6_t5_b2', 'cfg3_clevr_k7_t5_b1',
                                  'cfg5_clevr_k11_t7_b1'])
def test_train_step_matches_reference_goldens(case):
    g = load_golden(case)
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    loss = _train_step(m, x, eps)
    ref_loss = float(g['f32.train.loss'])
    assert abs(loss.item() - ref_loss) <= 1e-4 * abs(ref_loss)
    assert abs(loss.item() - float(g['f64.train.loss'])) <= 1e-4 * abs(ref_loss)
    assert rel_err(m.elbo_terms[:, 0].cpu(), g['f32.train.elbos']) < 1e-4
    bad = []
    for n, p in m.named_parameters():
        a = p.grad.double().cpu().flatten()
        ss, ref_ss = float((a * a).sum()), float(g[f'f64.train.grad.{n}.sumsq'])
        step = max(1, a.numel() // 16)
        smp = a[::step][:16].numpy()
        rms = np.sqrt(ref_ss / a.numel())
        if abs(ss - ref_ss) > 2e-3 * ref_ss + 1e-12 or np.abs(smp - g[f'f64.train.grad.{n}.sample']).max() > 2e-3 * rms + 1e-7:
            bad.append((n, ss, ref_ss))
    assert not bad, bad


@pytest.mark.parametrize('prec', [1, 0], ids=['split_f16x3', 'exact_fp32'])
@pytest.mark.parametrize('case', ['cfg3_clevr_k7_t5_b1', 'cfg5_clevr_k11_t7_b1'])
def test_headline_architecture_gradients_element_wise(case, prec):
    """Round 4: every gradient tensor of the HIP training step at the headline architecture against the FULL
    tensors of the reference's fp64 run (`<case>_grads.npz`, gen_goldens.py run_grad_case), element-wise: rel-L2 per tensor <= 1e-3
    (north_star gate), all tensors together <= 1e-3 - the same comparison tests/test_oracle_golden.py makes for the oracle."""
    from util import grad_views
    g, gg = load_golden(case), load_golden(case + '_grads')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params, options={'conv_precision': prec})       # round 5: the strict exact-fp32 path too (conv_precision 0)
    loss = _train_step(m, x, eps)
    assert abs(loss.item() - float(gg['f64.train.loss'])) <= 1e-4 * abs(float(gg['f64.train.loss']))
    assert rel_err(m.elbo_terms[:, 0].cpu(), gg['f64.train.elbos']) < 1e-4
    worst = max((rel_l2(*grad_views(n, p.grad.cpu().numpy(), gg['f64.train.grad.' + n])), n) for n, p in m.named_parameters())
    num = sum(float(((p.grad.double().cpu().numpy() - gg['f64.train.grad.' + n].astype(np.float64)) ** 2).sum()) for n, p in m.named_parameters())
    den = sum(float((gg['f64.train.grad.' + n].astype(np.float64) ** 2).sum()) for n, _ in m.named_parameters())
    print(f'[{case}, conv_precision {prec}] HIP vs reference fp64, element-wise: worst tensor {worst[1]} {worst[0]:.2e}, all tensors {np.sqrt(num / den):.2e}')
    assert worst[0] <= 1e-3, worst
    assert np.sqrt(num / den) <= 1e-3


def test_train_step_matches_oracle_on_fresh_inputs():
    from iodine_amd import synth
    arch = O.dsprites_arch(slots=3, iters=2)
    pn = synth.make_params(O.param_shapes(arch), seed=21, dec_gain=3.0, posterior_scale=0.05)
    params = {k: torch.from_numpy(v) for k, v in pn.items()}
    imgs, _ = synth.make_images(2, arch.img_size, seed=8, kind='blobs')
    x = torch.from_numpy(imgs)
    eps = torch.from_numpy(synth.make_eps(arch.iters, 2, arch.slots, arch.dim_latent, seed=10))
    out, grads = O.train_step_grads(x, eps, params, arch)
    m = make_hip_model(arch, params)
    loss = _train_step(m, x, eps)
    assert abs(loss.item() - out['loss'].item()) <= 1e-5 * abs(out['loss'].item())
    bad = [(n, rel_l2(p.grad.cpu().numpy(), grads[n].numpy())) for n, p in m.named_parameters()
           if not rel_l2(p.grad.cpu().numpy(), grads[n].numpy()) < 1e-3]
    assert not bad, bad


def test_backward_accumulates_and_scales_like_autograd():
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    _train_step(m, x, eps)
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    loss = m(x.to(DEV), eps.to(DEV))
    (0.5 * loss).backward()                          # no zero_grad: .grad accumulates (train.py:62 zeroes explicitly)
    for n, p in m.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), 1.5 * g1[n].cpu().numpy()) < 1e-5, n


def test_sgd_step_changes_next_loss():
    """set_params is re-run when the optimizer updates the weights in place (param version tracking)."""
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)                 # lib/solver/build.py:5-16
    l0 = _train_step(m, x, eps).item()
    opt.step()
    l1 = _train_step(m, x, eps).item()
    assert l1 != l0 and np.isfinite(l1)
    assert l1 < l0                                                   # one Adam step on the same batch lowers the loss


def test_gradients_form_one_flat_buffer():
    """after zero_grad(set_to_none) + backward every .grad is a slice of the one buffer iodine_train_backward filled, so
    the data-parallel all-reduce (iodine_amd.parallel.allreduce_gradients) runs on it in place, without gather/scatter"""
    from iodine_amd import parallel
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    m.zero_grad(set_to_none=True)
    m(x.to(DEV), eps.to(DEV)).backward()
    grads = [p.grad for p in m.parameters()]
    flat = parallel._shared_flat_view(grads)
    assert flat is not None and flat.numel() == sum(p.numel() for p in m.parameters())
    before = [gr.clone() for gr in grads]
    flat.mul_(2.0)                                                   # the view aliases the gradients
    assert all(torch.equal(gr, 2.0 * b) for gr, b in zip(grads, before))


def test_rccl_allreduce_on_flat_gradient_buffer():
    """RCCL (backend 'nccl') accepts the in-place flat view of the gradients: one-rank group on this GPU, SUM all-reduce
    leaves the values unchanged.  The two-rank arithmetic is covered on CPU by tests/test_parallel_cpu.py (gloo)."""
    import os
    import socket
    import torch.distributed as dist
    from iodine_amd import parallel
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    m.zero_grad(set_to_none=True)
    m(x.to(DEV), eps.to(DEV)).backward()
    grads = [p.grad for p in m.parameters()]
    before = [gr.clone() for gr in grads]
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        flat = parallel._shared_flat_view(grads)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(1)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(grads, before))
        parallel.allreduce_gradients(m.parameters())            # world 1: no-op path
        assert all(torch.equal(a, b) for a, b in zip(grads, before))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('prec', [1, 0], ids=['split_f16x3', 'exact_fp32'])
@pytest.mark.parametrize('name,ltol', [('traj_tiny', 2e-5), ('traj_cfg1', 5e-5)])
def test_training_trajectory_matches_reference(name, ltol, prec):
    """four steps of the reference's training-step body (lib/engine/train.py:58-65) with its optimizer (Adam, lr 3e-4,
    lib/solver/build.py:5-16): HIP forward/backward + the fused Adam kernel against the losses and final parameters the
    unmodified reference produced (tests/golden/gen_trajectory.py)"""
    from iodine_amd.optim import make_optimizer
    from util import check_trajectory_params, trajectory_setup
    tr, arch, params, x, eps = trajectory_setup(name)
    m = make_hip_model(arch, params, options={'conv_precision': prec})       # (the strict path: fp32-MFMA forms of every conv, round 5)
    opt = make_optimizer(m, base_lr=float(tr['meta_lr']), weight_decay=0.0)
    xd, losses = x.to(DEV), []
    for e in eps:
        loss = m(xd, e.to(DEV))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
    ref = tr['f64.losses']
    assert np.abs(np.array(losses) - ref).max() <= ltol * np.abs(ref).max(), (losses, list(ref))
    check_trajectory_params(tr, 'f32', m.named_parameters(), 0.05)
