"""Role timing of the warp-specialised tile conv (IODINE_CONV_PROF=1): python tools/v3_prof.py [epi]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['IODINE_CONV_PROF'] = '1'
from iodine_amd import _lib
L = _lib.lib()
N, S, C = 224, 128, 64
epi = int(sys.argv[1]) if len(sys.argv) > 1 else 0
w = (torch.rand(C, C, 3, 3) * 2 - 1).cuda() * 0.1
b = torch.zeros(C).cuda()
x = torch.rand(N, S, S, C, device='cuda') * 2 - 1
aux = torch.rand(N, S, S, C, device='cuda') * 2 - 1
out = torch.empty(N, S, S, C, device='cuda')
for _ in range(4):
    rc = L.iodine_op_conv3x3(None, 4, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(aux), _lib.ptr(out), N, S, S, C, C, C, C, 1, epi, epi)
    assert rc == 0
torch.cuda.synchronize()
