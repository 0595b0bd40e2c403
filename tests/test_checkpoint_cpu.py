"""SURVEY.md 8f-3 pin: a checkpoint WRITTEN BY THE REFERENCE (tests/golden/ckpt_tiny, made by
tests/golden/gen_checkpoint.py with the reference's Checkpointer, DataParallel wrapper and make_optimizer after two
steps of the traj_tiny trajectory) loads through iodine_amd.checkpoint into the module + optimizer pair, and the CPU
oracle continued from it reproduces steps 3 and 4 of the reference trajectory.  (The GPU continuation is in
tests/test_gpu_boundary.py.)"""
import os

import numpy as np
import torch

from iodine_amd import IODINE, checkpoint
from iodine_amd.optim import make_optimizer
from oracle import iodine_oracle as O
from util import GOLDEN, check_trajectory_params, hip_arch, trajectory_setup

CKPT_DIR = os.path.join(GOLDEN, 'ckpt_tiny')


def test_reference_written_checkpoint_loads_and_continues_the_reference_trajectory():
    tr, arch, params, x, eps = trajectory_setup('traj_tiny')
    path = checkpoint.last_checkpoint(CKPT_DIR)                                # reads the reference's checkpoint.pkl index
    assert path is not None and os.path.basename(path) == 'model_0002.pth'
    raw = torch.load(path, map_location='cpu')
    assert set(raw) == {'model', 'optimizer', 'epoch'} and raw['epoch'] == 2
    assert all(k.startswith('module.') for k in raw['model'])                  # DataParallel wrapper, checkpoint.py:43
    assert len(raw['optimizer']['param_groups']) == len(params)                # one group per parameter, solver/build.py:10-14

    m = IODINE(hip_arch(arch))                                                 # module on CPU: parameters only, no compute
    opt = make_optimizer(m, base_lr=float(tr['meta_lr']))
    extra = checkpoint.load_checkpoint(path, m, opt)
    assert extra == {'epoch': 2}
    assert [n for n, _ in m.named_parameters()] == list(params.keys())
    for p in m.parameters():
        st = opt.state[p]
        assert int(st['step']) == 2 and st['exp_avg'].shape == p.shape and st['exp_avg_sq'].shape == p.shape

    # continue with the CPU oracle + torch.optim.Adam built from OUR optimizer's state_dict (what the reference's
    # Checkpointer.load would hand to its own optimizer): steps 3 and 4 of the reference trajectory
    ps = {n: torch.nn.Parameter(p.detach().clone()) for n, p in m.named_parameters()}
    ref_opt = torch.optim.Adam([{'params': [p], 'lr': float(tr['meta_lr']), 'weight_decay': 0.0} for p in ps.values()],
                               lr=float(tr['meta_lr']))
    ref_opt.load_state_dict(opt.state_dict())
    for s in (2, 3):
        out, grads = O.train_step_grads(x, eps[s], {k: v.detach() for k, v in ps.items()}, arch)
        assert abs(out['loss'].item() - float(tr['f32.losses'][s])) <= 2e-5 * abs(float(tr['f32.losses'][s])), s
        for n, p in ps.items():
            p.grad = grads[n]
        ref_opt.step()
    check_trajectory_params(tr, 'f32', ps.items(), tol=0.05)


def test_checkpoint_written_here_has_the_reference_layout(tmp_path):
    _, arch, params, _, _ = trajectory_setup('traj_tiny')
    m = IODINE(hip_arch(arch))
    m.load_state_dict(params)
    opt = make_optimizer(m, base_lr=3e-4)
    raw_ref = torch.load(checkpoint.last_checkpoint(CKPT_DIR), map_location='cpu')
    opt.load_state_dict(raw_ref['optimizer'])
    path = str(tmp_path / 'model_0003.pth')
    checkpoint.save_checkpoint(path, m, opt, data_parallel_prefix=True, epoch=3)
    mine = torch.load(path, map_location='cpu')
    assert list(mine['model'].keys()) == list(raw_ref['model'].keys())
    assert [g.keys() for g in mine['optimizer']['param_groups']] == [g.keys() for g in raw_ref['optimizer']['param_groups']]
    assert [g['params'] for g in mine['optimizer']['param_groups']] == [g['params'] for g in raw_ref['optimizer']['param_groups']]
    for i, st in raw_ref['optimizer']['state'].items():
        assert set(mine['optimizer']['state'][i]) == set(st)
        assert torch.equal(mine['optimizer']['state'][i]['exp_avg'], st['exp_avg'])
        assert float(mine['optimizer']['state'][i]['step']) == float(st['step'])
