import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import iodine_oracle as O
from util import load_golden, golden_setup, make_hip_model
g = load_golden('tiny_k5'); arch, params, x, eps, _ = golden_setup(g)
B, K = x.shape[0], arch.slots
trace = []
ref = O.reconstruct(x, eps, params, arch, trace=trace)
m = make_hip_model(arch, params)
m.set_option('stop_after_iters', 1)
m.reconstruct(x.cuda(), eps.cuda())
print('elbo it0', m.elbo_terms.cpu().numpy(), float(trace[0]['elbo']))
z = m.debug_buffer('z', 0).cpu().view(B, K, -1); print('z err', float((z - trace[0]['z']).abs().max()))
dec = m.debug_buffer('dec_out', 0).cpu().view(B * K, -1, 4)
mean = torch.sigmoid(dec[..., :3]).view(B, K, arch.img_size, arch.img_size, 3).permute(0, 1, 4, 2, 3)
print('mean err', float((mean - trace[0]['mean']).abs().max()), 'logit err', float((dec[..., 3].view(B, K, 1, arch.img_size, arch.img_size) - trace[0]['logits']).abs().max()))
for l in range(arch.dec_layers):
    a = m.debug_buffer(f'act{l}', 0).cpu()
    print('act', l, float(a.abs().max()), bool(torch.isfinite(a).all()))
for k in ('g_pm', 'g_plv'):
    v = m.debug_buffer(k, 0).cpu().view(B, K, -1); print(k, 'err', float((v - trace[0][k]).abs().max()), float(trace[0][k].abs().max()))
enc = m.debug_buffer('enc', 0).cpu().view(B, K, -1, 20)
print('enc shape ok; ract0', float(m.debug_buffer('ract0', 0).abs().max()))
print('pm after it0 err', float((m.debug_buffer('pm', 0).cpu().view(B, K, -1) - trace[1]['post_mean']).abs().max()) if len(trace) > 1 else None)
