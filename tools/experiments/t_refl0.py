"""refine_l0_fused on / off: bitwise-level agreement of the refinement activations and the step (cfg3 golden), timing per category"""
import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import golden_setup, load_golden, make_hip_model, rel_l2, rel_err
g = load_golden('cfg3_clevr_k7_t5_b1')
arch, params, x, eps, _ = golden_setup(g)
res = {}
for v in (0, 1):
    m = make_hip_model(arch, params)
    m.set_option('refine_l0_fused', v)
    m.zero_grad(set_to_none=True)
    loss = m(x.cuda(), eps.cuda()); loss.backward(); torch.cuda.synchronize()
    res[v] = (loss.item(), m.elbo_terms[:, 0].cpu().clone(), {n: p.grad.cpu().numpy().copy() for n, p in m.named_parameters()})
print('loss', res[0][0], res[1][0], 'elbo rel', rel_err(res[1][1], res[0][1]))
num = sum(float(((res[1][2][n].astype(np.float64) - res[0][2][n]) ** 2).sum()) for n in res[0][2]); den = sum(float((res[0][2][n].astype(np.float64) ** 2).sum()) for n in res[0][2])
print('grad rel-L2 fused l0 vs split', (num / den) ** 0.5)
print('vs golden loss', abs(res[1][0] - float(g['f32.train.loss'])) / abs(float(g['f32.train.loss'])))
# inference: reconstruct outputs fused vs split
out = {}
for v in (0, 1):
    m = make_hip_model(arch, params)
    m.set_option('refine_l0_fused', v)
    r = m.reconstruct(x.cuda(), eps.cuda()); torch.cuda.synchronize()
    out[v] = {k: (t.cpu().clone() if torch.is_tensor(t) else t) for k, t in (r.items() if isinstance(r, dict) else enumerate(r))}
for k in out[0]:
    if torch.is_tensor(out[0][k]): print('reconstruct', k, rel_err(out[1][k], out[0][k]))
