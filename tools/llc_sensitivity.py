"""Does the 64->64 tile conv speed up when its input and output fit in the 256 MiB Infinity Cache?  Same kernel, same
data distribution, N slot-images per launch; prints ns per tile-slot (512 resident blocks) so tail effects are visible."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iodine_amd import _lib
L = _lib.lib()
S, C = 128, 64
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = (torch.rand(C, C, 3, 3) * 2 - 1).cuda() * 0.1
b = torch.zeros(C).cuda()
for N in (8, 16, 24, 32, 48, 64, 96, 128, 224):
    x = torch.rand(N, S, S, C, device='cuda') * 2 - 1
    out = torch.empty(N, S, S, C, device='cuda')
    def run():
        rc = L.iodine_op_conv3x3(None, MODE, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(out), N, S, S, C, C, C, C, 1, 0, 0)
        assert rc == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    reps = max(4, 800 // N)
    t0 = time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tiles = N * 64
    print(f'N={N:4d} in+out {2 * N * S * S * C * 4 / 2**20:7.0f} MiB  {dt * 1e3:7.3f} ms/launch (incl. weight pack + sync)  {dt * 1e6 / tiles * 512:7.2f} us per 512 tiles', flush=True)
