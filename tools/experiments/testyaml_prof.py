"""Training step of the reference's configs/test.yaml architecture (REF 32 ch x 3 layers KERNEL_SIZE 5, DEC 32 ch x 5 layers KERNEL_SIZE 5, 64 px,
K = 6, T = 5, L = 16, ENCODING subset; batch 32): the configuration whose REFINEMENT stack runs on the generic stride-2 kernels.
    rocprofv3 --kernel-trace --stats -- python tools/experiments/testyaml_prof.py [batch]"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
import torch
from iodine_amd import IODINE, synth
from iodine_amd.model import arch_namespace
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
a = arch_namespace(16, 5, 6, 64, (32, 3, 128), (32, 5), sigma=0.14, kernels=(5, 5),
                   encoding=['posterior', 'grad_post', 'image', 'leave_one_out_likelihood'])
m = IODINE(a)
sh = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(sh, seed=0).items()})
m = m.to('cuda:0'); m.manual_seed(7)
x = torch.from_numpy(synth.make_images(B, 64, seed=0)).cuda()
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.zero_grad(set_to_none=True); m(x).backward()
    torch.cuda.synchronize(); print('train ms', (time.perf_counter() - t0) * 1e3, flush=True)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.reconstruct(x)
    torch.cuda.synchronize(); print('reconstruct ms', (time.perf_counter() - t0) * 1e3, flush=True)
