"""Is a configuration's step host-bound?  Enqueue time of n steps (host clock before the final synchronize) against their wall time.
    python tools/experiments/host_bound.py [clevr6|dsprites] [train|infer] [n]"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
import torch
from iodine_amd import IODINE, synth
from iodine_amd.model import clevr6_arch, dsprites_arch
from iodine_amd.optim import make_optimizer
cfg = sys.argv[1] if len(sys.argv) > 1 else 'dsprites'
mode = sys.argv[2] if len(sys.argv) > 2 else 'train'
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
arch = clevr6_arch() if cfg == 'clevr6' else dsprites_arch()
m = IODINE(arch).to('cuda:0'); m.manual_seed(7)
x = torch.from_numpy(synth.make_images(32, arch.IMG_SIZE, seed=0, kind='uniform')).cuda()
opt = make_optimizer(m, base_lr=3e-4, weight_decay=0.0)
def step():
    if mode == 'infer':
        m.reconstruct(x); return
    loss = m(x); m.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{cfg} {mode}: enqueue {1e3 * (t1 - t0) / n:.3f} ms / step, wall {1e3 * (t2 - t0) / n:.3f} ms / step, drain after the last enqueue {1e3 * (t2 - t1):.2f} ms')
