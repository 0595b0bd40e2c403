#!/usr/bin/env python3
"""Golden TRAINING TRAJECTORIES from the unmodified reference (build container only): the body of
lib/engine/train.py:58-65 (``loss = model(data).mean(); optimizer.zero_grad(); loss.backward(); optimizer.step()``)
repeated for a few steps with the reference's optimizer (``torch.optim.Adam(lr=BASE_LR, weight_decay=WEIGHT_DECAY)``,
lib/solver/build.py:5-16 with configs/clevr6_prop.yaml:19-20), same images every step, a fresh epsilon stream per step
(seed SEED_E + step).  Stored: the loss of every step (fp32 and fp64 runs) and the parameters after the last step.

Usage:  python tests/golden/gen_trajectory.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_goldens as G  # noqa: E402  (puts the repo root and /root/reference on sys.path, imports the reference)
import torch  # noqa: E402
from iodine_amd import synth  # noqa: E402

STEPS, LR = 4, 3e-4
TRAJ = {'traj_tiny': 'tiny', 'traj_cfg1': 'cfg1_dsprites_k4_t3_b4'}


def run(name, case):
    fam, K, T, B, kind = G.CASES[case]
    S, L = G.ARCHS[fam]['S'], G.ARCHS[fam]['L']
    imgs = synth.make_images(B, S, seed=G.SEED_X, kind=kind)
    imgs = imgs[0] if kind == 'blobs' else imgs
    out = dict(meta_case=case, meta_steps=STEPS, meta_lr=LR, meta_eps_seed0=G.SEED_E)
    for tag, dtype in (('f32', torch.float32), ('f64', torch.float64)):
        model, _ = G.build_reference(fam, K, T, dtype)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=0.0)
        x = torch.from_numpy(imgs).to(dtype)
        losses = []
        for s in range(STEPS):
            e = torch.from_numpy(synth.make_eps(T, B, K, L, seed=G.SEED_E + s)).to(dtype)
            with G.EpsReplay(e) as rp:
                loss = model(x).mean()
                assert rp.i == T + 1
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        out[f'{tag}.losses'] = np.array(losses, dtype=np.float64)
        for n, p in model.named_parameters():
            if case == 'tiny':
                out[f'{tag}.param.{n}'] = p.detach().double().numpy().copy()
            else:
                G.summarize(f'{tag}.param.{n}', p, out)
        print(f'  [{name}/{tag}] losses', ' '.join(f'{v:.4f}' for v in losses))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    for name, case in TRAJ.items():
        run(name, case)
