"""Which kernel family owns the error on a hard trained-weights case?  Runs tests/golden/teacher_cfg1 step 20 with the mask logits
sharpened x8 under several kernel options against the CPU oracle (fp32 and fp64)."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from iodine_amd import synth
from oracle import iodine_oracle as O
from util import load_golden, make_hip_model
name, ckpt, k = sys.argv[1] if len(sys.argv) > 1 else 'teacher_cfg1', int(sys.argv[2]) if len(sys.argv) > 2 else 20, float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
t = load_golden(name)
fam, K, T, B = str(t['meta_family']), int(t['meta_K']), int(t['meta_T']), int(t['meta_B'])
arch = {'tiny': O.tiny_arch, 'dsprites': O.dsprites_arch}[fam](slots=K, iters=T)
sw, sx, se = (int(v) for v in t['meta_seeds'])
params = {kk: torch.from_numpy(t[f'ckpt{ckpt}.param.{kk}']).clone() for kk in O.param_shapes(arch)}
params['decoder.conv.weight'][3] *= k; params['decoder.conv.bias'][3] *= k
imgs, _ = synth.make_images(B, arch.img_size, seed=sx, kind='blobs')
x = torch.from_numpy(imgs)
eps = torch.from_numpy(synth.make_eps(T, B, K, arch.dim_latent, seed=se + 5000 + ckpt))
trace = []
o64 = O.train_forward(x.double(), eps.double(), {a: b.double() for a, b in params.items()}, arch, trace)
print('fp64 elbos', o64['elbos'].detach().numpy())
for opts in ({}, {'conv_precision': 0}, {'conv_variant': 1}, {'refine_split': 0}, {'fuse_l0': 0}):
    m = make_hip_model(arch, params)
    for kk, v in opts.items():
        m.set_option(kk, v)
    loss = m(x.cuda(), eps.cuda())
    e = m.elbo_terms[:, 0].double().cpu() - o64['elbos'].detach()
    print(opts, 'elbo err per iteration', e.numpy(), ' kl', (m.elbo_terms[:, 1].double().cpu()).numpy())
    # stage-by-stage on iteration 0..: decoder output, g
    if not opts:
        for it in range(T + 1):
            z = m.debug_buffer('z', it).cpu().view(B, K, -1).double()
            print(' iter', it, 'z err', float((z - trace[it]['z']).abs().max()) if it < len(trace) else None)

# ---- first refinement layer, iteration 0: split vs unsplit vs fp64 conv of the oracle's encoding ----
import torch.nn.functional as F
enc64 = trace[0]['enc'].double().flatten(0, 1)          # (B*K, 17, S, S)
w0, b0 = params['refine.mlc.layers.0.weight'].double(), params['refine.mlc.layers.0.bias'].double()
ref0 = F.elu(F.conv2d(enc64, w0, b0, stride=2, padding=1)).permute(0, 2, 3, 1).contiguous()
print('enc channel max |.|:', enc64.abs().amax((0, 2, 3)).numpy().round(2))
for split in (1, 0):
    m = make_hip_model(arch, params)
    m.set_option('refine_split', split)
    m.set_option('stop_after_iters', 1)
    m.reconstruct(x.cuda(), eps.cuda())
    enc = m.debug_buffer('enc', 0).cpu().view(B * K, -1, 20)[..., :17].double()
    e_enc = (enc - enc64.permute(0, 2, 3, 1).reshape(B * K, -1, 17)).abs().amax((0, 1))
    r0 = m.debug_buffer('ract0', 0).cpu().view(ref0.shape).double()
    err = (r0 - ref0).abs()
    own = F.elu(F.conv2d(enc.view(B * K, arch.img_size, arch.img_size, 17).permute(0, 3, 1, 2), w0, b0, stride=2, padding=1)).permute(0, 2, 3, 1)
    print(f'   conv of the module\'s OWN encoding in fp64 vs its layer-0 output: max err {float((r0 - own).abs().max()):.3e}; own enc ch14 max |.| {float(enc[..., 14].abs().max()):.1f}, ch7 {float(enc[..., 7].abs().max()):.1f}')
    print(f'refine_split={split}: enc max err per channel', e_enc.numpy().round(7), ' layer-0 output: max err', float(err.max()), 'max |ref|', float(ref0.abs().max()),
          ' worst pixel', np.unravel_index(int(err.argmax()), err.shape))

# ---- where does the split form's 100-sigma leave-one-out value come from? ----
print('--- leave-one-out channel, iteration 0 ---')
encs_ = {}
for split in (1, 0):
    m = make_hip_model(arch, params)
    m.set_option('refine_split', split); m.set_option('stop_after_iters', 1)
    m.reconstruct(x.cuda(), eps.cuda())
    encs_[split] = m.debug_buffer('enc', 0).cpu().view(B, K, -1, 20)
    lnstat = m.debug_buffer('lnstat', 0).cpu().view(B, K, 8)
    print(f'split={split} lnstat[loo mean, inv std] of image 0:', lnstat[0, :, 4:6].numpy())
d = (encs_[1][..., 14] - encs_[0][..., 14]).abs()
idx = np.unravel_index(int(d.argmax()), d.shape)
print('max |split - unsplit| in ch14:', float(d.max()), 'at (b,k,p)', idx, ' values', float(encs_[1][idx][14]), float(encs_[0][idx][14]),
      ' oracle fp64', float(enc64.view(B, K, 17, -1)[idx[0], idx[1], 14, idx[2]]))
print('all 17 channels at that pixel, split:', encs_[1][idx][:17].numpy().round(4))
print('all 17 channels at that pixel, unsplit:', encs_[0][idx][:17].numpy().round(4))
print('number of pixels where ch14 differs by > 1e-3:', int((d > 1e-3).sum()), 'of', d.numel(), '; other channels max diff:', float((encs_[1][..., :14] - encs_[0][..., :14]).abs().max()))
