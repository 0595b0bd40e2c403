"""Interleaved same-process A/B of several builds of the library (iodine_amd/ab/libws_*.so, tools/ws_variants.sh build ...):
every round runs the 64->64 conv (forward, then data gradient) once per variant, so all variants see the same clock / thermal
state.  Run under `rocprofv3 --kernel-trace --output-format csv`; tools/ws_ab_report.py maps the dispatch order back to variants."""
import ctypes as C, glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iodine_amd import _lib
_lib.lib()
libs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'iodine_amd', 'ab', 'libws_*.so')))
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
epi = int(os.environ.get('WS_AB_EPI', 1))     # data-gradient form timed as the second launch: 1 stored, 4 / 5 row sums
N, S, Cc = int(os.environ.get('N', 224)), int(os.environ.get('S', 128)), int(os.environ.get('C', 64))
w = (torch.rand(Cc, Cc, 3, 3) * 2 - 1).cuda() * 0.1
b = torch.zeros(Cc).cuda()
out = torch.empty(N, S, S, Cc, device='cuda')
x = torch.rand(N, S, S, Cc, device='cuda') * 2 - 1
a = torch.rand(N, S, S, Cc, device='cuda') * 2 - 1
Ls = []
for p in libs:
    L = C.CDLL(p)
    L.iodine_op_conv3x3.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 10
    Ls.append(L)
print('VARIANTS', ' '.join(os.path.basename(p)[6:-3] for p in libs), flush=True)
for r in range(rounds):
    for L in Ls:
        assert L.iodine_op_conv3x3(None, mode, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(out), N, S, S, Cc, Cc, Cc, Cc, 1, 0, 0) == 0
        assert L.iodine_op_conv3x3(None, mode, _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(a), _lib.ptr(out), N, S, S, Cc, Cc, Cc, Cc, 1, epi, 1) == 0
torch.cuda.synchronize()
