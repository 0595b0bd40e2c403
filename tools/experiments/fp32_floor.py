"""The reference-precision floor on the sharpened trained-weights cases: the CPU oracle in fp32 against itself in fp64 (the same
comparison tests/test_gpu_trained_weights.py makes between the HIP path and the fp32 oracle).  CPU only."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import iodine_oracle as O
from iodine_amd import synth
from util import rel_err, rel_l2
torch.set_num_threads(8)
t = np.load('tests/golden/teacher_cfg1_long.npz')
arch = O.dsprites_arch(slots=4, iters=3)
sw, sx, se = (int(v) for v in t['meta_seeds'])
imgs, _ = synth.make_images(4, arch.img_size, seed=sx, kind='blobs')
cases = [(3000, 8.0), (3000, 32.0), (3000, 128.0)]
for c, sh in cases:
    res = {}
    for dt in (torch.float32, torch.float64):
        params = {k: torch.from_numpy(t[f'ckpt{c}.param.{k}']).clone() for k in O.param_shapes(arch)}
        params['decoder.conv.weight'][3] *= sh; params['decoder.conv.bias'][3] *= sh
        params = {k: v.to(dt) for k, v in params.items()}
        x = torch.from_numpy(imgs).to(dt)
        eps = torch.from_numpy(synth.make_eps(3, 4, 4, arch.dim_latent, seed=se + 5000 + c)).to(dt)
        out, rg = O.train_step_grads(x, eps, params, arch)
        rec = O.reconstruct(x, eps, params, arch)
        res[dt] = (out, rg, rec)
    o32, g32, r32 = res[torch.float32]; o64, g64, r64 = res[torch.float64]
    scale = max(abs(float(o64['loss'])), float(o64['elbos'].abs().max()))
    num = sum(float(((g32[n].double() - g64[n]) ** 2).sum()) for n in g64); den = sum(float((g64[n] ** 2).sum()) for n in g64)
    worst = max((rel_l2(g32[n].numpy(), g64[n].numpy()), n) for n in g64 if float(g64[n].abs().max()) > 0 and n != 'decoder.conv.bias')
    print(f'ckpt {c} x{sh:g}: oracle fp32 vs fp64: loss {abs(float(o32["loss"]) - float(o64["loss"])) / scale:.1e}, '
          f'elbos {float((o32["elbos"].double() - o64["elbos"]).abs().max()) / scale:.1e}, grad rel-L2 {(num / den) ** .5:.1e}, '
          f'worst {worst[1]} {worst[0]:.1e}, recon pred {rel_err(r32["pred"], r64["pred"]):.1e}, recon elbos {rel_err(r32["elbos"], r64["elbos"]):.1e}, '
          f'argmax agree {float((r32["mask"][:, :, 0].argmax(1) == r64["mask"][:, :, 0].argmax(1)).float().mean()):.5f}', flush=True)
