"""What does each group of helper kernels cost END TO END?  Needs a library built with -DIODINE_XSKIP_HOOK
(IODINE_EXTRA_HIPCC_FLAGS=-DIODINE_XSKIP_HOOK python -m iodine_amd.build after touching the sources; load it through
IODINE_HIP_LIB): for every group the step is timed with the group's launches turned into no-ops and compared with the
full step measured right before and after.  Results of the ablated steps are wrong by construction: timing only.
usage: python tools/helper_cost.py [train|infer]"""
import sys, time
import torch
sys.path.insert(0, '.')
from iodine_amd import IODINE, synth
from iodine_amd.model import clevr6_arch

mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
GROUPS = {1: 'partial-tile reductions (fold / reduce / colsum)', 2: 'head-backward GEMMs (sgemm)', 4: 'refine head',
          8: 'pixel passes', 16: 'broadcast layer forward', 32: 'broadcast layer backward', 64: 'pointwise / axpy / transpose',
          128: 'output conv forward', 256: 'output conv data gradient', 512: 'refinement convs forward',
          1024: "refinement conv gradients (dgrad only: wgrad needs its reduction)", 2048: 'output conv weight gradient'}
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


def timeit(n=5):
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


timeit()
for bit, name in GROUPS.items():
    m.set_option('xskip', 0); a = timeit()
    m.set_option('xskip', bit); s = timeit()
    m.set_option('xskip', 0); b = timeit()
    print(f'{name:48s} full {0.5 * (a + b):7.3f} ms   without {s:7.3f} ms   cost {0.5 * (a + b) - s:+6.3f} ms', flush=True)
