"""A/B timing of one library option on the SAME box (box-to-box spread is ~2 %, larger than most single optimisations).
usage: python tools/ab_bench.py <option> <valueA> <valueB> [train|infer] [reps]"""
import sys, time
import torch
sys.path.insert(0, '.')
from iodine_amd import IODINE, synth
from iodine_amd.model import clevr6_arch

opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else 'train'
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
B = 32
arch = clevr6_arch()
m = IODINE(arch).to('cuda:0')
x = torch.from_numpy(synth.make_images(B, 128, seed=0, kind='uniform')).cuda()
eps = torch.from_numpy(synth.make_eps(arch.ITERS, B, arch.SLOTS, arch.DIM_LATENT, seed=1)).cuda()


def step():
    if mode == 'train':
        m.zero_grad(set_to_none=True)
        m(x, eps).backward()
    else:
        m.reconstruct(x, eps)


def timeit(n=6):
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {va: [], vb: []}
for r in range(reps):
    for v in (va, vb):
        m.set_option(opt, v)
        res[v].append(timeit())
for v in (va, vb):
    t = sorted(res[v])
    print(f'{opt}={v}: median {t[len(t) // 2]:.3f} ms  min {t[0]:.3f}  max {t[-1]:.3f}  ({mode}, B={B})')
