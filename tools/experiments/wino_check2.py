"""Per-shape report for the Winograd experiment (no asserts): accuracy of forward / data gradient vs fp64 and run-to-run bitwise equality."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.nn.functional as F
from test_gpu_ops import _conv_op, _rand, nhwc, rel_err
C_ = 64
for S, N in [(16, 1), (32, 3), (64, 5), (128, 2), (16, 300), (128, 32)]:
    x = _rand(N, C_, S, S, seed=21); w = _rand(C_, C_, 3, 3, seed=22, scale=3.0 / (C_ * 9) ** 0.5); b = _rand(C_, seed=23, scale=0.5)
    ref = nhwc(F.elu(F.conv2d(x.double(), w.double(), b.double(), padding=1))).float()
    f = [_conv_op(11, nhwc(x), w, b, None, N, S, S, C_, C_, C_, C_, 1, 0, 0, ref.shape) for _ in range(3)]
    g = _rand(N, C_, S, S, seed=24, scale=1e-3); g[N // 2:] *= 1e-4
    a = F.elu(_rand(N, C_, S, S, seed=25, scale=2.0))
    refd = nhwc((F.conv_transpose2d(g.double(), w.double(), padding=1) * torch.where(a > 0, torch.ones_like(a), a + 1).double()).float())
    d = [_conv_op(11, nhwc(g), w, None, nhwc(a), N, S, S, C_, C_, C_, C_, 1, 1, 1, refd.shape) for _ in range(3)]
    nf = [int((f[0] != f[i]).sum()) for i in (1, 2)]; nd = [int((d[0] != d[i]).sum()) for i in (1, 2)]
    ed = max(rel_err(d[0][h], refd[h]) for h in ((slice(0, N // 2), slice(N // 2, N)) if N > 1 else (slice(0, 1),)))
    msg = f'S={S} N={N}: fwd err {rel_err(f[0], ref):.2e} differing elements between runs {nf} | dgrad err {ed:.2e} differing {nd}'
    if nd[0]:
        idx = (d[0] != d[1]).nonzero()[:6].tolist()
        msg += f' first diffs (n,y,x,c) {idx}'
    if nf[0]:
        idx = (f[0] != f[1]).nonzero()[:6].tolist()
        msg += f' fwd diffs {idx}'
    print(msg, flush=True)
