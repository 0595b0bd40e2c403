#!/usr/bin/env python3
"""A checkpoint WRITTEN BY THE REFERENCE (build container only): ``Checkpointer.save`` of lib/utils/checkpoint.py:36-54
on the tiny-architecture reference model wrapped in ``torch.nn.DataParallel`` (lib/modeling/build.py:11-12, so the
state_dict keys carry the ``module.`` prefix, checkpoint.py:43) with the reference's optimizer (``make_optimizer`` of
lib/solver/build.py:5-16: Adam, one param group per parameter) after the first TWO steps of the ``traj_tiny``
trajectory (gen_trajectory.py: same images, epsilon seed SEED_E + step).  Output: tests/golden/ckpt_tiny/model_0002.pth
and the ``checkpoint.pkl`` index ``Checkpointer`` keeps next to it.  Continuing from it must reproduce steps 3 and 4 of
traj_tiny.npz (tests/test_gpu_extras.py, tests/test_checkpoint_cpu.py).

Usage:  python tests/golden/gen_checkpoint.py
"""
import os
import shutil
import sys
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_goldens as G  # noqa: E402  (repo root + /root/reference on sys.path; imports the reference model)
import torch  # noqa: E402
from iodine_amd import synth  # noqa: E402
from lib.solver.build import make_optimizer as ref_make_optimizer  # noqa: E402  (the reference)
from lib.utils.checkpoint import Checkpointer as RefCheckpointer  # noqa: E402  (the reference)

STEPS_BEFORE_SAVE, LR = 2, 3e-4


def main():
    fam, K, T, B, kind = G.CASES['tiny']
    S, L = G.ARCHS[fam]['S'], G.ARCHS[fam]['L']
    model, _ = G.build_reference(fam, K, T, torch.float32)
    model = torch.nn.DataParallel(model)            # CPU: a pass-through wrapper, but state_dict keys get 'module.'
    model.train()
    cfg = SimpleNamespace(TRAIN=SimpleNamespace(BASE_LR=LR, WEIGHT_DECAY=0.0))
    opt = ref_make_optimizer(cfg, model)
    x = torch.from_numpy(synth.make_images(B, S, seed=G.SEED_X, kind=kind))
    losses = []
    for s in range(STEPS_BEFORE_SAVE):
        e = torch.from_numpy(synth.make_eps(T, B, K, L, seed=G.SEED_E + s))
        with G.EpsReplay(e):
            loss = model(x).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    out = os.path.join(HERE, 'ckpt_tiny')
    shutil.rmtree(out, ignore_errors=True)
    ck = RefCheckpointer(model, opt, args={'epoch': 0}, save_dir=out)
    ck.args['epoch'] = STEPS_BEFORE_SAVE
    ck.save('model_{:04d}'.format(STEPS_BEFORE_SAVE))
    print('losses before the save', losses)
    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)), 'bytes')


if __name__ == '__main__':
    main()
