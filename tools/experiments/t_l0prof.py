"""phase counters of the fused encoding + layer-0 kernel (library built with -DIODINE_TILE_PROF): one cfg3 reconstruct"""
import sys, torch
sys.path.insert(0, '.')
from iodine_amd import IODINE, synth
from iodine_amd.model import clevr6_arch
arch = clevr6_arch(); B = 32
m = IODINE(arch).to('cuda:0')
x = torch.from_numpy(synth.make_images(B, 128, seed=0, kind='uniform')).cuda()
eps = torch.from_numpy(synth.make_eps(arch.ITERS, B, arch.SLOTS, arch.DIM_LATENT, seed=1)).cuda()
m.set_option('graph', 0)
m.reconstruct(x, eps); torch.cuda.synchronize()
