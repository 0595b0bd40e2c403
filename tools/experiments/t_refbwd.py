import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import golden_setup, load_golden, make_hip_model, rel_l2
g = load_golden('cfg3_clevr_k7_t5_b1')
arch, params, x, eps, _ = golden_setup(g)
res = {}
for v in (0, 1):
    m = make_hip_model(arch, params)
    m.set_option('refine_bwd_fused', v)
    m.zero_grad(set_to_none=True)
    loss = m(x.cuda(), eps.cuda()); loss.backward(); torch.cuda.synchronize()
    res[v] = {n: p.grad.cpu().numpy().copy() for n, p in m.named_parameters()}
for n in res[0]:
    if 'refine.mlc.layers.0' in n or 'refine.mlc.layers.1' in n:
        print(n, 'fused vs two-kernel rel_l2', rel_l2(res[1][n], res[0][n]), 'max', np.abs(res[0][n]).max())
gg = load_golden('cfg3_clevr_k7_t5_b1_grads')
for v in (0, 1):
    for n in ('refine.mlc.layers.0.weight', 'refine.mlc.layers.0.bias'):
        print(v, n, 'vs reference fp64', rel_l2(res[v][n], gg['f64.train.grad.' + n]))
