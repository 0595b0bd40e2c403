"""Time the 64->64 3x3 conv kernels at the cfg3 shape (N=224, S=128) under `rocprofv3 --kernel-trace`: forward and data
gradient of every op-level mode given on the command line, on random data.  (kernel durations come from the trace)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iodine_amd import _lib
L = _lib.lib()
N, S, C = int(os.environ.get('N', 224)), int(os.environ.get('S', 128)), int(os.environ.get('C', 64))
w = (torch.rand(C, C, 3, 3) * 2 - 1).cuda() * 0.1
b = torch.zeros(C).cuda()
out = torch.empty(N, S, S, C, device='cuda')
x = torch.rand(N, S, S, C, device='cuda') * 2 - 1
a = torch.rand(N, S, S, C, device='cuda') * 2 - 1
for mode in [int(v) for v in sys.argv[1:]] or [2, 9, 10]:
    for rep in range(6):
        assert L.iodine_op_conv3x3(None, mode, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(out), N, S, S, C, C, C, C, 1, 0, 0) == 0
    for rep in range(6):
        assert L.iodine_op_conv3x3(None, mode, _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(a), _lib.ptr(out), N, S, S, C, C, C, C, 1, 1, 1) == 0
    torch.cuda.synchronize()
    print('done', mode, flush=True)
