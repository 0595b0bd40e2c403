"""Op-level check of the Winograd experiment kernel (tools/experiments/kernels_wino.hip) against fp64 and against the product's
weight-stationary kernel.  Build the side library first (tools/wino_variants.sh default), then on the GPU box:
    IODINE_HIP_LIB=$PWD/iodine_amd/ab/libwino_default.so python tools/experiments/wino_check.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.nn.functional as F
from test_gpu_ops import _conv_op, _rand, nhwc, rel_err


def check(S, N):
    """forward + bias + ELU and the data gradient x ELU' against fp64, incl. small-magnitude gradients and tiles of very different
    ranges in one launch; deterministic; same arithmetic class as the weight-stationary direct kernel (op mode 10)"""
    C_ = 64
    x = _rand(N, C_, S, S, seed=21)
    w = _rand(C_, C_, 3, 3, seed=22, scale=3.0 / (C_ * 9) ** 0.5)
    b = _rand(C_, seed=23, scale=0.5)
    ref = nhwc(F.elu(F.conv2d(x.double(), w.double(), b.double(), padding=1))).float()
    got = _conv_op(11, nhwc(x), w, b, None, N, S, S, C_, C_, C_, C_, 1, 0, 0, ref.shape)
    e_f = rel_err(got, ref)
    assert e_f < 3e-6, e_f
    g = _rand(N, C_, S, S, seed=24, scale=1e-3)                  # small-magnitude gradients
    g[N // 2:] *= 1e-4                                           # ... and tiles with very different ranges in one launch
    a = F.elu(_rand(N, C_, S, S, seed=25, scale=2.0))
    refd = nhwc((F.conv_transpose2d(g.double(), w.double(), padding=1) * torch.where(a > 0, torch.ones_like(a), a + 1).double()).float())
    gotd = _conv_op(11, nhwc(g), w, None, nhwc(a), N, S, S, C_, C_, C_, C_, 1, 1, 1, refd.shape)
    e_d = 0.0
    for half in ((slice(0, N // 2), slice(N // 2, N)) if N > 1 else (slice(0, 1),)):
        e_d = max(e_d, rel_err(gotd[half], refd[half]))
    assert e_d < 3e-6, e_d
    assert torch.equal(gotd, _conv_op(11, nhwc(g), w, None, nhwc(a), N, S, S, C_, C_, C_, C_, 1, 1, 1, refd.shape))   # deterministic
    e_ws = rel_err(_conv_op(10, nhwc(x), w, b, None, N, S, S, C_, C_, C_, C_, 1, 0, 0, ref.shape), ref)
    return e_f, e_d, e_ws


if __name__ == '__main__':
    for S, N in [(32, 3), (128, 2), (16, 300), (64, 5), (16, 1), (128, 32)]:
        e_f, e_d, e_ws = check(S, N)
        print(f'ok S={S} N={N}: max err / max |ref|  forward {e_f:.2e}  data gradient {e_d:.2e}  (weight-stationary direct kernel, forward: {e_ws:.2e})', flush=True)
