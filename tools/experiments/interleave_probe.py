"""Does the ORDER of dense (MFMA-bound, power-limited) and memory-bound launches matter?  Six 64->64 conv launches at the cfg3 shape and six
1.9 GB device copies, back to back in blocks (DDDDDD MMMMMM) or interleaved (DM DM DM ...), total time by events, many repetitions."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from iodine_amd import _lib
L = _lib.lib()
N, S, C = 224, 128, 64
w = (torch.rand(C, C, 3, 3) * 2 - 1).cuda() * 0.1
b = torch.zeros(C).cuda()
out = torch.empty(N, S, S, C, device='cuda')
x = torch.rand(N, S, S, C, device='cuda') * 2 - 1
src = torch.rand(N, S, S, C, device='cuda')
dst = torch.empty_like(src)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def dense():
    assert L.iodine_op_conv3x3(None, mode, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(out), N, S, S, C, C, C, C, 1, 0, 0) == 0


def mem():
    dst.copy_(src)


def run(seq, reps=8):
    for f in seq:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for f in seq:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(3):
    t_d = run([dense] * 6)
    t_m = run([mem] * 6)
    t_block = run([dense] * 6 + [mem] * 6)
    t_inter = run([dense, mem] * 6)
    t_inter3 = run(([dense] * 3 + [mem] * 3) * 2)
    print(f'dense x6 {t_d:.3f} ms  mem x6 {t_m:.3f} ms  sum {t_d + t_m:.3f} | DDDDDDMMMMMM {t_block:.3f}  DDDMMMDDDMMM {t_inter3:.3f}  DMDMDM.. {t_inter:.3f}', flush=True)
