#!/usr/bin/env python3
"""Benchmark of the IODINE refinement step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--mode infer|train] [--config clevr6|dsprites]

A "step" is one pass of the hot path over one synthetic batch resident in HBM:
  * mode infer: ``model.reconstruct(x)`` = T refinement iterations + final decode
    (lib/modeling/iodine.py:107-112, the step lib/eval/ari_eval.py:22 times)
  * mode train: ``loss = model(x); loss.backward()`` (lib/engine/train.py:60-63, no optimizer)
Metric: image-refinement-iterations/s = (images in the job) * T / t_step, whole job over all ranks.
Workload at N=1: BASELINE.json configs[2] (headline): CLEVR6 128x128, K=7, T=5, batch 32; for N>1 every rank
runs its own 32 images (configs[3] at N=8), no data-path collective in inference ("weak" scaling).

Extra objects on the JSON line: ``roofline`` for the dominant kernel (fp32-MFMA 3x3 conv 64->64, duration from
HIP events recorded on the launch stream inside the timed region) and ``cpu_baseline`` (the CPU oracle timed on
this box's host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from iodine_amd import IODINE, parallel, synth  # noqa: E402
from iodine_amd.model import clevr6_arch, dsprites_arch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (the 5 PF headline is 2:1 sparse)
SPLIT_PASSES = 3                   # fp32 product = a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, three f16 MFMAs, fp32 accumulate
PUBLISHED_TRAIN_ITERS_PER_S = 94.0  # 1.7 s / 32-image T=5 training step on 4 unknown GPUs (log.md:3)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--mode', choices=['infer', 'train'], default=os.environ.get('IODINE_BENCH_MODE', 'train'))
    ap.add_argument('--config', choices=['clevr6', 'dsprites'], default='clevr6')
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--slots', type=int, default=None)
    ap.add_argument('--iters', type=int, default=None)
    ap.add_argument('--conv-precision', type=int, choices=[0, 1], default=1,
                    help='decoder 3x3 convs: 1 = fp32 operands split into fp16 hi+lo, 3 f16 MFMAs (default); 0 = exact fp32 MFMA')
    ap.add_argument('--no-adam', action='store_true', help='time forward + backward only (default: + fused Adam step, '
                    'like the reference batch_time, lib/engine/train.py:58-67)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=None)
    return ap.parse_args()


def build_model(args, device):
    if args.config == 'clevr6':
        arch = clevr6_arch(slots=args.slots or 7, iters=args.iters or 5)
    else:
        arch = dsprites_arch(slots=args.slots or 6, iters=args.iters or 5)
    model = IODINE(arch)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    params = synth.make_params(shapes, seed=0)              # torch-default-init bounds, deterministic bytes
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return model.to(device), arch, params


def cpu_baseline(args, arch, params, mode):
    """Time the CPU oracle (PyTorch-CPU restatement of the reference) on a bounded sample of the same workload."""
    from oracle import iodine_oracle as O
    from util import hip_arch  # noqa: F401
    oa = O.Arch(dim_latent=arch.DIM_LATENT, iters=arch.ITERS, slots=arch.SLOTS, sigma=arch.SIGMA,
                img_size=arch.IMG_SIZE, ref_chan=arch.REF.CONV_CHAN, ref_layers=arch.REF.CONV_LAYERS,
                ref_mlp=arch.REF.MLP_UNITS, dec_chan=arch.DEC.CONV_CHAN, dec_layers=arch.DEC.CONV_LAYERS)
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    Bc = args.cpu_batch or (2 if args.config == 'clevr6' else 8)
    p = {k: torch.from_numpy(v) for k, v in params.items()}
    x = torch.from_numpy(synth.make_images(Bc, oa.img_size, seed=0))
    eps = torch.from_numpy(synth.make_eps(oa.iters, Bc, oa.slots, oa.dim_latent, seed=1))
    fn = (lambda: O.reconstruct(x, eps, p, oa)) if mode == 'infer' else (lambda: O.train_step_grads(x, eps, p, oa))
    fn()                                                    # warm-up
    reps, t0 = 0, time.perf_counter()
    while reps < 2 or (time.perf_counter() - t0 < 10.0 and reps < 8):
        out = fn()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    ref_elbos = (out['elbos'] if mode == 'infer' else out[0]['elbos']).detach().double().numpy()
    return dict(value=Bc * oa.iters / dt, unit='image-refinement-iters/s', cores=threads, kind='port',
                sample=f'{mode} step, batch {Bc} of the same workload, {reps} reps after 1 warm-up, '
                       f'{dt * 1e3:.0f} ms/step; oracle/iodine_oracle.py (PyTorch-CPU fp32)'), (x, eps, ref_elbos)


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)'
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)                 # before the process group: RCCL binds the communicator to this device
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    model, arch, params = build_model(args, device)
    B, T, K, S = args.batch, arch.ITERS, arch.SLOTS, arch.IMG_SIZE
    x = torch.from_numpy(synth.make_images(B, S, seed=0, first_index=rank * B)).to(device)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    model.generator = gen
    eps = torch.randn((T + 1, B, K, arch.DIM_LATENT), device=device, generator=gen)

    if args.mode == 'infer':
        def step():
            return model.reconstruct(x, eps)
    else:
        from iodine_amd.optim import make_optimizer
        opt = None if args.no_adam else make_optimizer(model, base_lr=3e-4, weight_decay=0.0)   # configs/clevr6_prop.yaml:19-20

        def step():
            model.zero_grad(set_to_none=True)
            loss = model(x, eps)
            loss.backward()
            parallel.allreduce_gradients(model.parameters(), world)       # one RCCL all-reduce of the flat grads
            if opt is not None:
                opt.step()
            return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    model.set_option('conv_precision', args.conv_precision)
    for _ in range(args.warmup):
        step()
    model.set_option('profile', 1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    model.set_option('profile', 0)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * T / (dt / args.steps)

    # secondary measurement (not `value`): the other step type on the same workload
    other = None
    if args.mode == 'train':
        def istep():
            return model.reconstruct(x, eps)
        istep()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            istep()
        barrier()
        dti = (time.perf_counter() - t1) / args.steps
        if world > 1:
            t = torch.tensor([dti], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dti = float(t.item())
        other = dict(step='infer (reconstruct: T iterations + final decode)', ms_per_step=round(dti * 1e3, 3),
                     image_refinement_iters_per_s=round(world * B * T / dti, 2))

    # ---- roofline of the dominant kernel: conv3x3_tile_kernel<C,C,*> (decoder 3x3 conv C->C, fwd + dgrad) ----
    # Inside the timed region only the dominant launches were bracketed with events (profile level 1: every category
    # costs ~1 ms per step in event records); the per-category table comes from two extra, untimed steps at level 2.
    C_ = arch.DEC.CONV_CHAN
    flops_per_launch = 2.0 * C_ * C_ * 9 * S * S * B * K
    CATS = ('conv_tile_fwd', 'conv_tile_dgrad', 'conv_tile_wgrad', 'dec_out', 'dec_out_dgrad', 'dec_out_wgrad',
            'dec_l0', 'l0_reduce', 'l0_slot_sum', 'pixel_pass1', 'pixel_pass2', 'refine_conv', 'refine_head',
            'refine_wgrad', 'refine_dgrad', 'refine_bias_grad')

    def read_prof():
        out = {}
        for cat in CATS:
            tot, cnt = model.profile_read(cat)
            if cnt:
                out[cat] = dict(ms_total=round(tot, 3), launches=cnt, ms_avg=round(tot / cnt, 4))
        return out

    prof = read_prof()                                   # timed region: conv_tile_* only
    model.set_option('profile', 2)
    for _ in range(2):
        step()
    barrier()
    model.set_option('profile', 0)
    prof_all = read_prof()
    for cat, v in prof_all.items():
        prof.setdefault(cat, dict(v, note='untimed pass'))
    dom_ms = sum(prof[c]['ms_total'] for c in ('conv_tile_fwd', 'conv_tile_dgrad', 'conv_tile_wgrad') if c in prof)
    dom_n = sum(prof[c]['launches'] for c in ('conv_tile_fwd', 'conv_tile_dgrad', 'conv_tile_wgrad') if c in prof)
    achieved = flops_per_launch / (dom_ms / dom_n * 1e-3) / 1e12 if dom_n else 0.0
    if args.conv_precision == 1:
        # the kernel executes SPLIT_PASSES f16 MFMAs per algorithmic (fp32-class) multiply-add: the peak for ALGORITHMIC
        # FLOPs is the dense f16 MFMA peak divided by the number of passes
        peak = PEAK_F16_MFMA_TFLOPS / SPLIT_PASSES
        kname = (f'conv3x3_tile_f16x3_kernel<{C_},{C_}> + conv3x3_wgrad_f16x3_ws_kernel<{C_},{C_}> (decoder 3x3 conv {C_}->{C_}: '
                 f'fwd, dgrad, wgrad launches; fp32 in/out, operands split into f16 hi+lo, 3 f16 MFMAs, fp32 accumulate)')
    else:
        peak = PEAK_F32_MFMA_TFLOPS
        kname = (f'conv3x3_tile_kernel<{C_},{C_}> + conv3x3_wgrad_tile_kernel<{C_},{C_}> (decoder 3x3 conv {C_}->{C_}: '
                 f'fwd, dgrad, wgrad launches; exact fp32 MFMA)')
    # HBM traffic per launch from the rocprofv3 PMC pass recorded in profiles/ (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
    # measured on the forward launch of this shape; null for other shapes
    traffic = 1.06e9 + 0.94e9 if (args.config == 'clevr6' and B == 32 and K == 7 and args.conv_precision == 1) else None
    roofline = dict(bound='mfma', kernel=kname, achieved=round(achieved, 2), peak=round(peak, 1), unit='TFLOP/s',
                    frac=round(achieved / peak, 4), traffic=traffic,
                    flops_per_launch=flops_per_launch, avg_launch_ms=round(dom_ms / max(dom_n, 1), 4),
                    launches=dom_n, kernel_time_share=round(dom_ms / (dt * 1e3), 4),
                    executed_mfma_tflops=round(achieved * (SPLIT_PASSES if args.conv_precision == 1 else 1), 1),
                    algorithmic_hbm_bytes_per_launch=2.0 * B * K * S * S * C_ * 4,
                    hbm_gbps_algorithmic=round(2.0 * B * K * S * S * C_ * 4 / (dom_ms / max(dom_n, 1) * 1e-3) / 1e9, 1))

    out = dict(metric='refinement_iters_per_s', value=round(value, 2), unit='image-refinement-iters/s',
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3),
               higher_is_better=True, scaling='weak',
               vs_baseline=(round(value / PUBLISHED_TRAIN_ITERS_PER_S, 3) if args.mode == 'train' and args.config == 'clevr6' else None),
               dtype=('f32 (3xf16-split MFMA convs, f32 accumulate)' if args.conv_precision == 1 else 'f32'), data='synthetic',
               config=dict(workload=f'{"CLEVR6 128x128" if args.config == "clevr6" else "multi-dSprites 64x64"}, '
                                    f'K={K}, T={T}, batch {B}/GPU, {args.mode} step '
                                    f'({"reconstruct: T iterations + final decode" if args.mode == "infer" else ("forward + backward" + ("" if args.no_adam else " + fused Adam step"))})',
                           step=args.mode, global_batch=B * world, slots=K, iters=T, img_size=S,
                           parallelism=f'dp{world} (images sharded over ranks; '
                                       f'{"one RCCL all-reduce of the flat gradient buffer per step" if args.mode == "train" else "no data-path collective"})'),
               batch_iters_per_s=round(T / (dt / args.steps), 3), roofline=roofline, kernels=prof)
    if other:
        out['inference_step'] = other

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, (xc, ec, ref_elbos) = cpu_baseline(args, arch, params, args.mode)
        out['cpu_baseline'] = cb
        # parity of this very run against the oracle on the CPU sample (gate 1e-3, north_star); the timed training
        # steps moved the weights (Adam), so the initial ones - what the oracle was given - are loaded back first
        with torch.no_grad():
            model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        if args.mode == 'infer':
            model.reconstruct(xc.to(device), ec.to(device))
        else:
            model(xc.to(device), ec.to(device))
        got = model.elbo_terms[:, 0].double().cpu().numpy()
        n = min(len(got), len(ref_elbos))
        out['elbo_rel_err_vs_cpu'] = float(abs(got[:n] - ref_elbos[:n]).max() / abs(ref_elbos[:n]).max())
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
