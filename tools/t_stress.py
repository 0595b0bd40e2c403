import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from test_gpu_fullsize import _full_batch
from util import make_hip_model
nbad = 0
for case, B, reps in (('cfg5_clevr_k11_t7_b1', 8, 40), ('cfg3_clevr_k7_t5_b1', 32, 15), ('cfg2_dsprites_k6_t5_b2', 32, 40)):
    g, arch, params, x, eps = _full_batch(case, B)
    m = make_hip_model(arch, params)
    xd, ed = x.cuda(), eps.cuda()
    ref = None
    for rep in range(reps):
        out = m.reconstruct(xd, ed)
        m.zero_grad(set_to_none=True)
        loss = m(xd, ed); loss.backward()
        cur = [t.clone() for t in out] + [loss.detach().clone()] + [p.grad.clone() for p in m.parameters()]
        if ref is None: ref = cur
        else:
            bad = [i for i, (a, b) in enumerate(zip(cur, ref)) if not torch.equal(a, b)]
            if bad:
                nbad += 1
                print(case, 'rep', rep, 'DIFFERS in', bad[:6], 'max diff', max(float((cur[i] - ref[i]).abs().max()) for i in bad), flush=True)
    print(case, 'done', reps, flush=True)
print('total differing reps', nbad)
