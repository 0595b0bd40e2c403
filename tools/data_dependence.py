"""Does the conv kernel's duration depend on the DATA (same instruction stream)?  If all-zero / constant inputs run
much faster than random ones the kernel is power/current-limited, not issue-limited.  Run under rocprofv3 --kernel-trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iodine_amd import _lib
L = _lib.lib()
N, S, C = 224, 128, 64
w = (torch.rand(C, C, 3, 3) * 2 - 1).cuda() * 0.1
b = torch.zeros(C).cuda()
out = torch.empty(N, S, S, C, device='cuda')
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 2          # 2 = tile kernel, 4 = warp-specialised variant
for name, x in (('random', (torch.rand(N, S, S, C, device='cuda') * 2 - 1)), ('zeros', torch.zeros(N, S, S, C, device='cuda')),
                ('const', torch.full((N, S, S, C), 0.37, device='cuda'))):
    for _ in range(8):
        rc = L.iodine_op_conv3x3(None, MODE, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(out), N, S, S, C, C, C, C, 1, 0, 0)
        assert rc == 0
    torch.cuda.synchronize()
    print('done', name, flush=True)
