"""Same-process, interleaved A/B of the Winograd kernel (op mode 11) against the weight-stationary direct kernel (mode 10) on the
cfg3 launch shape (N = 224 slot-images of 128 x 128 x 64, random data): every round runs forward + data gradient of each, so both
see the same clock / thermal state.  Run under `rocprofv3 --kernel-trace --stats` (tools/wino_ab.sh) - the op entry point packs
the weights and computes the per-cell max on every call, so host timing would not isolate the conv kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iodine_amd import _lib
L = _lib.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
modes = [int(m) for m in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['10', '11'])]
N, S, Cc = int(os.environ.get('N', 224)), int(os.environ.get('S', 128)), 64
kind = os.environ.get('DATA', 'random')
w = (torch.rand(Cc, Cc, 3, 3) * 2 - 1).cuda() * 0.1
b = torch.zeros(Cc).cuda()
out = torch.empty(N, S, S, Cc, device='cuda')
x = (torch.rand(N, S, S, Cc, device='cuda') * 2 - 1) if kind == 'random' else torch.zeros(N, S, S, Cc, device='cuda')
a = torch.rand(N, S, S, Cc, device='cuda') * 2 - 1
for r in range(rounds):
    for m in modes:
        assert L.iodine_op_conv3x3(None, m, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(out), N, S, S, Cc, Cc, Cc, Cc, 1, 0, 0) == 0
        assert L.iodine_op_conv3x3(None, m, _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(a), _lib.ptr(out), N, S, S, Cc, Cc, Cc, Cc, 1, 1, 1) == 0
torch.cuda.synchronize()
