"""Phase profile of the exact-fp32 weight-stationary conv (op mode 12) at the cfg3 shape; library built with -DIODINE_TILE_PROF
(IODINE_HIP_LIB=...): python tools/experiments/ws32f_prof.py [C S N]"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from iodine_amd import _lib
L = _lib.lib()
C, S, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 128, 224)
x = torch.rand(N, S, S, C, device='cuda') * 2 - 1
w = (torch.rand(C, C, 3, 3, device='cuda') * 2 - 1) * 0.1
b = torch.zeros(C, device='cuda')
out = torch.empty(N, S, S, C, device='cuda')
for epi, aux in ((0, None), (1, x)):
    for i in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.iodine_op_conv3x3(None, 12, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(aux), _lib.ptr(out), N, S, S, C, C, C, C, 1, epi, epi)
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
        print('epi', epi, 'op wall ms (incl. pack + sync)', e0.elapsed_time(e1), flush=True)
