"""SURVEY.md section 8f rows next to the hot path: fused Adam, device-side ARI tables, reference checkpoint format."""
import numpy as np
import pytest
import torch

from oracle import ari_oracle as A
from oracle import iodine_oracle as O
from util import golden_setup, load_golden, make_hip_model, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('wd', [0.0, 0.01])
def test_fused_adam_matches_torch_adam(wd):
    from iodine_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 17, 3, 3), (64,), (1024, 512), (7,), (4, 64, 3, 3)]
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    hip_p = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]
    ref_opt = torch.optim.Adam(ref_p, lr=3e-4, weight_decay=wd)            # lib/solver/build.py:5-16
    hip_opt = FusedAdam(hip_p, lr=3e-4, weight_decay=wd)
    for it in range(5):
        for rp, hp in zip(ref_p, hip_p):
            gr = torch.randn(rp.shape, generator=g) * (10.0 ** (it - 2))
            rp.grad = gr.clone()
            hp.grad = gr.clone().to(DEV)
        ref_opt.step()
        hip_opt.step()
        for rp, hp in zip(ref_p, hip_p):
            assert rel_err(hp.detach().cpu(), rp.detach()) < 2e-6, it
    st = hip_opt.state[hip_p[0]]
    assert st['step'] == 5 and rel_err(st['exp_avg_sq'].cpu(), ref_opt.state[ref_p[0]]['exp_avg_sq']) < 1e-5


def test_ari_tables_match_oracle_and_golden():
    from iodine_amd.ari import ARIEvaluator, ari_tables, compute_ari
    g = load_golden('cfg1_dsprites_k4_t3_b4')
    arch, params, x, eps, gt = golden_setup(g)
    m = make_hip_model(arch, params)
    pred, mask, mean = m.reconstruct(x.to(DEV), eps.to(DEV))
    tables = ari_tables(mask, gt)
    onehot = A.binarize_argmax(mask.cpu().numpy())
    for b in range(len(gt)):
        ref_table = A.contingency(gt[b], onehot[b])
        assert np.array_equal(tables[b, :gt[b].shape[0]], ref_table)                  # integer work: bit-exact
        assert abs(compute_ari(tables[b, :gt[b].shape[0]]) - float(g['f32.recon.ari'][b])) <= 1e-3
    gk = load_golden('ari')
    assert abs(compute_ari(gk['known.table']) - float(gk['known.ari'])) < 1e-12      # lib/utils/ari.py:56-63
    assert compute_ari(gk['perfect.table']) == 1.0

    class Replay:                                   # evaluator protocol of lib/eval/ari_eval.py:13-39
        def reconstruct(self, image):
            return m.reconstruct(image, eps.to(DEV))
    ev = ARIEvaluator()
    ev.evaluate(Replay(), (x.to(DEV), gt))
    assert len(ev.aris) == len(gt) and np.abs(np.array(ev.aris) - g['f32.recon.ari']).max() <= 1e-3
    assert ev.get_results().startswith('Ari: ')


def test_reference_checkpoint_format_roundtrip(tmp_path):
    from iodine_amd import checkpoint
    from iodine_amd.optim import make_optimizer
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    m = make_hip_model(arch, params)
    opt = make_optimizer(m, base_lr=3e-4)
    m.zero_grad(set_to_none=True)
    m(x.to(DEV), eps.to(DEV)).backward()
    opt.step()
    path = str(tmp_path / 'model_0001.pth')
    checkpoint.save_checkpoint(path, m, opt, data_parallel_prefix=True, epoch=1, iter=17)     # checkpoint.py:36-54
    raw = torch.load(path, map_location='cpu')
    assert set(raw) == {'model', 'optimizer', 'epoch', 'iter'} and all(k.startswith('module.') for k in raw['model'])
    m2 = make_hip_model(arch, {k: torch.zeros_like(v) for k, v in params.items()})
    opt2 = make_optimizer(m2, base_lr=3e-4)
    extra = checkpoint.load_checkpoint(path, m2, opt2)
    assert extra == {'epoch': 1, 'iter': 17}
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1.detach().cpu(), p2.detach().cpu())
    # the restored pair continues identically
    for mm, oo in ((m, opt), (m2, opt2)):
        mm.zero_grad(set_to_none=True)
        mm(x.to(DEV), eps.to(DEV)).backward()
        oo.step()
    for (n1, p1), p2 in zip(m.named_parameters(), m2.parameters()):
        assert rel_err(p2.detach().cpu(), p1.detach().cpu()) < 1e-6, (n1, rel_err(p2.detach().cpu(), p1.detach().cpu()))


def test_reference_shaped_train_and_eval_loops(tmp_path):
    """lib/engine/train.py:44-108 / eval.py:14-28 shaped loops on synthetic scenes: the loss goes down, the checkpoint
    round-trips, the ARI evaluator runs under no_grad."""
    from iodine_amd import IODINE, engine
    from iodine_amd.data import make_dataloader
    from iodine_amd.model import arch_namespace
    from iodine_amd.optim import make_optimizer
    torch.manual_seed(0)
    m = IODINE(arch_namespace(8, 2, 3, 16, (32, 2, 32), (32, 2))).to(DEV)
    opt = make_optimizer(m, base_lr=3e-3)
    ds = engine.SyntheticScenes(8, 16)
    dl = make_dataloader(ds, batch_size=4, shuffle=False)
    ck = str(tmp_path / 'model.pth')
    losses = engine.train(m, opt, dl, torch.device(DEV), max_steps=30, print_every=1000, checkpoint_path=ck)
    assert len(losses) == 30 and np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5])
    ev = engine.evaluate(m, dl, torch.device(DEV))
    assert len(ev.aris) == 8 and all(-1.0 <= a <= 1.0 for a in ev.aris)
    m2 = IODINE(arch_namespace(8, 2, 3, 16, (32, 2, 32), (32, 2))).to(DEV)
    from iodine_amd.checkpoint import load_checkpoint
    load_checkpoint(ck, m2)
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
