"""Phase profile of the stride-1 weight-gradient kernel (build with IODINE_EXTRA_HIPCC_FLAGS=-DIODINE_TILE_PROF)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iodine_amd import _lib
L = _lib.lib()
N, S, C = 224, 128, 64
x = (torch.rand(N, S, S, C, device='cuda') * 2 - 1)
d = (torch.rand(N, S, S, C, device='cuda') * 2 - 1) * 1e-3
gw = torch.zeros(C, C, 3, 3, device='cuda'); gb = torch.zeros(C, device='cuda')
for _ in range(4):
    rc = L.iodine_op_conv3x3_wgrad(None, _lib.ptr(x), _lib.ptr(d), _lib.ptr(gw), _lib.ptr(gb), N, S, C, C, C, 1)
    assert rc == 0
torch.cuda.synchronize()
