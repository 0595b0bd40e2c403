import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch, torch.nn.functional as F
from iodine_amd import _lib
from util import nhwc, rel_err
L = _lib.lib()
def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
def conv_op(mode, x, w, b, aux, N, S, C, epi, tflip, shape):
    out = torch.full(shape, float('nan'), device='cuda')
    xs, ws = x.cuda().contiguous(), w.cuda().contiguous()
    bs = b.cuda() if b is not None else None
    ax = aux.cuda().contiguous() if aux is not None else None
    rc = L.iodine_op_conv3x3(None, mode, _lib.ptr(xs), _lib.ptr(ws), _lib.ptr(bs), _lib.ptr(ax), _lib.ptr(out), N, S, S, C, C, C, C, 1, epi, tflip)
    assert rc == 0, L.iodine_last_error(None)
    torch.cuda.synchronize()
    return out.cpu()
for (C, S, N) in [(64, 32, 3), (32, 16, 2), (64, 128, 2), (64, 16, 70), (32, 64, 9)]:
    x = rnd(N, C, S, S, seed=21); w = rnd(C, C, 3, 3, seed=22, scale=3.0 / (C * 9) ** 0.5); b = rnd(C, seed=23, scale=0.5)
    ref = nhwc(F.elu(F.conv2d(x.double(), w.double(), b.double(), padding=1))).float()
    for mode in (10,):
        got = conv_op(mode, nhwc(x), w, b, None, N, S, C, 0, 0, ref.shape)
        print('fwd', C, S, N, mode, rel_err(got, ref), flush=True)
    g = rnd(N, C, S, S, seed=24, scale=1e-3); a = F.elu(rnd(N, C, S, S, seed=25, scale=2.0))
    refd = nhwc((F.conv_transpose2d(g.double(), w.double(), padding=1) * torch.where(a > 0, torch.ones_like(a), a + 1).double()).float())
    for mode in (10,):
        gotd = conv_op(mode, nhwc(g), w, None, nhwc(a), N, S, C, 1, 1, refd.shape)
        print('dgrad', C, S, N, mode, rel_err(gotd, refd), flush=True)
        tiles = S // 16
        rows = conv_op(mode, nhwc(g), w, None, nhwc(a), N, S, C, 4, 1, (N, S, tiles, 3, C))
        r = refd.view(N, S, tiles, 16, C).double()
        want = torch.zeros(N, S, tiles, 3, C, dtype=torch.float64)
        want[:, :, :, 1] = r.sum(3)
        want[:, :, 0, 0] = r[:, :, 0, 0]; want[:, :, 0, 1] -= r[:, :, 0, 0]
        want[:, :, -1, 2] = r[:, :, -1, 15]; want[:, :, -1, 1] -= r[:, :, -1, 15]
        print('l0rows', C, S, N, mode, rel_err(rows, want.float()), flush=True)
