import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from iodine_amd import _lib
L = _lib.lib()
def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
N, S, C = 88, 128, 64
x = rnd(N, S, S, C, seed=1).cuda(); w = rnd(C, C, 3, 3, seed=2, scale=0.1).cuda(); b = rnd(C, seed=3).cuda(); a = rnd(N, S, S, C, seed=4).cuda()
for epi, tflip, shape in ((0, 0, (N, S, S, C)), (1, 1, (N, S, S, C)), (4, 1, (N, S, S // 16, 3, C))):
    outs = []
    for rep in range(6):
        out = torch.full(shape, float('nan'), device='cuda')
        rc = L.iodine_op_conv3x3(None, 10, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b) if epi == 0 else None, _lib.ptr(a) if epi else None, _lib.ptr(out), N, S, S, C, C, C, C, 1, epi, tflip)
        assert rc == 0
        torch.cuda.synchronize()
        outs.append(out.clone())
    for i in range(1, 6):
        d = (outs[i] != outs[0])
        print('epi', epi, 'rep', i, 'differs:', int(d.sum().item()), 'nan:', int(torch.isnan(outs[i]).sum().item()),
              'where n:', sorted(set(d.nonzero()[:, 0].tolist()))[:8] if d.any() else '', flush=True)
