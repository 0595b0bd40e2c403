"""Same-box A/B of library builds, each step type: python tools/ab_libs.py [--rounds R] [--modes train,infer] [--config clevr6|dsprites|cfg5] libA.so libB.so ...
Every round runs each build once in its own process (IODINE_HIP_LIB), alternating, so that all builds see the same box and the same
thermal state; prints the median / min over rounds (box-to-box spread is ~2 %, larger than most single optimisations)."""
import os
import subprocess
import sys

args = sys.argv[1:]
rounds, modes, config = 3, ['train', 'infer'], 'clevr6'
while args and args[0].startswith('--'):
    k = args.pop(0)
    v = args.pop(0)
    if k == '--rounds':
        rounds = int(v)
    elif k == '--modes':
        modes = v.split(',')
    elif k == '--config':
        config = v
libs = args
code = r'''
import sys, time, torch
sys.path.insert(0, '.')
from iodine_amd import IODINE, synth
from iodine_amd.model import clevr6_arch, dsprites_arch
mode, config = sys.argv[1], sys.argv[2]
arch, B = {'clevr6': (clevr6_arch(), 32), 'dsprites': (dsprites_arch(), 32), 'cfg5': (clevr6_arch(slots=11, iters=7), 8)}[config]
m = IODINE(arch).to('cuda:0')
x = torch.from_numpy(synth.make_images(B, arch.IMG_SIZE, seed=0, kind='uniform')).cuda()
eps = torch.from_numpy(synth.make_eps(arch.ITERS, B, arch.SLOTS, arch.DIM_LATENT, seed=1)).cuda()
def step():
    if mode == 'train':
        m.zero_grad(set_to_none=True); m(x, eps).backward()
    else:
        m.reconstruct(x, eps)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10 if config != 'dsprites' else 50
for _ in range(n): step()
torch.cuda.synchronize()
print((time.perf_counter() - t0) / n * 1e3)
'''
res = {(l, m): [] for l in libs for m in modes}
for r in range(rounds):
    for m in modes:
        for l in libs:
            env = dict(os.environ, IODINE_HIP_LIB=os.path.abspath(l))
            out = subprocess.run([sys.executable, '-c', code, m, config], env=env, capture_output=True, text=True)
            try:
                res[(l, m)].append(float(out.stdout.strip().splitlines()[-1]))
            except Exception:
                print('FAILED', l, m, out.stderr[-800:])
for m in modes:
    for l in libs:
        t = sorted(res[(l, m)])
        if t:
            print(f'{m:6s} {os.path.basename(l):28s} median {t[len(t) // 2]:8.3f} ms   min {t[0]:8.3f}   max {t[-1]:8.3f}   ({len(t)} rounds, {config}, fwd+bwd no Adam)')
