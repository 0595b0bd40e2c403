"""Event-measured average launch time of every kernel category on the cfg3 step:
python tools/cat_times.py [train|infer] [option=value ...]"""
import sys
import torch
sys.path.insert(0, '.')
from iodine_amd import IODINE, synth
from iodine_amd.model import clevr6_arch

mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
B = 32
arch = clevr6_arch()
m = IODINE(arch).to('cuda:0')
for kv in sys.argv[2:]:
    k, v = kv.split('=')
    m.set_option(k, float(v))
x = torch.from_numpy(synth.make_images(B, 128, seed=0, kind='uniform')).cuda()
eps = torch.from_numpy(synth.make_eps(arch.ITERS, B, arch.SLOTS, arch.DIM_LATENT, seed=1)).cuda()


def step():
    if mode == 'train':
        m.zero_grad(set_to_none=True)
        m(x, eps).backward()
    else:
        m.reconstruct(x, eps)


for _ in range(2):
    step()
torch.cuda.synchronize()
m.set_option('profile', 2)
N = 4
for _ in range(N):
    step()
torch.cuda.synchronize()
m.set_option('profile', 0)
tot = 0.0
for cat in ('conv_tile_fwd', 'conv_tile_dgrad', 'conv_tile_wgrad', 'dec_out', 'dec_out_dgrad', 'dec_out_wgrad', 'dec_out_bwd', 'dec_l0',
            'l0_reduce', 'l0_slot_sum', 'pixel_pass1', 'pixel_pass2', 'refine_l0', 'refine_l0f', 'refine_conv', 'refine_head', 'refine_wgrad',
            'refine_dgrad', 'refine_bwd01', 'refine_bias_grad'):
    ms, cnt = m.profile_read(cat)
    if cnt:
        tot += ms / N
        print(f'{cat:18s} {cnt // N:4d} launches/step  avg {ms / cnt * 1e3:8.1f} us   {ms / N:7.3f} ms/step')
print(f'sum of categories {tot:.3f} ms/step ({mode}; event brackets add ~1 ms to the step)')
