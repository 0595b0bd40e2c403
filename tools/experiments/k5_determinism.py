"""generic decoder (CLEVR shapes with DEC.KERNEL_SIZE 5, batch 4): two runs of N training steps from the same seed must agree bit for bit
(fixed-order sums everywhere) and stay finite"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
import torch
from iodine_amd import IODINE, synth
from iodine_amd.model import arch_namespace

def run(nsteps):
    a5 = arch_namespace(64, 5, 7, 128, (64, 4, 256), (64, 4), kernels=(3, 5))
    m = IODINE(a5)
    sh = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(sh, seed=0).items()})
    m = m.to('cuda:0'); m.manual_seed(7)
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    x = torch.from_numpy(synth.make_images(4, 128, seed=0)).cuda()
    losses = []
    for i in range(nsteps):
        opt.zero_grad(set_to_none=True)
        loss = m(x); loss.backward(); opt.step()
        losses.append(loss.item())
    return losses, [p.detach().clone() for p in m.parameters()]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
l1, p1 = run(n); l2, p2 = run(n)
print('losses', l1[0], l1[-1], 'finite', all(map(lambda v: v == v and abs(v) < 1e30, l1)))
print('bitwise equal losses', l1 == l2, 'params', all(torch.equal(a, b) for a, b in zip(p1, p2)))
