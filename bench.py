#!/usr/bin/env python3
"""Benchmark of the IODINE refinement step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--mode infer|train] [--config clevr6|dsprites] [--graph 0|1]

A "step" is one pass of the hot path over one synthetic batch resident in HBM:
  * mode train (default): ``loss = model(x); optimizer.zero_grad(); loss.backward(); optimizer.step()`` = the timed body of
    lib/engine/train.py:58-65, i.e. forward + backward + the (fused) Adam step, with the T+1 reparameterisation-noise draws
    of Gaussian.sample (iodine.py:632) inside the step like the reference's batch_time (the library's own Philox generator);
    ``--no-adam`` times forward + backward only;
  * mode infer: ``model.reconstruct(x)`` = T refinement iterations + final decode (lib/modeling/iodine.py:107-112, the step
    lib/eval/ari_eval.py:22 times).
Metric: image-refinement-iterations/s = (images in the job) * T / t_step, whole job over all ranks.
Workload at N=1: BASELINE.json configs[2] (headline): CLEVR6 128x128, K=7, T=5, batch 32; ``--config dsprites`` is configs[1]
(multi-dSprites 64x64, K=6, T=5, batch 32).  For N>1 every rank runs its own 32 images (configs[3] at N=8): "weak" scaling,
one RCCL all-reduce of the flat gradient buffer per training step, no data-path collective in inference.

``--gpus N`` with N>1 and no launcher environment starts its own N ranks (``torch.distributed.run``, one process per GPU);
under the driver's ``python -m torch.distributed.run ... bench.py --gpus N`` the ranks are used as given.

Extra objects on the JSON line: ``roofline`` for the dominant kernel (split-fp16 MFMA 3x3 conv C->C; duration from HIP events
recorded on the launch stream inside the timed region; ``traffic`` from the tracked PMC file profiles/r02_pmc.json while its
source digest still matches the tree), ``cpu_baseline`` (the CPU oracle timed on this box's host cores on a bounded sample,
rank 0 at N=1 only) with ELBO and gradient parity of this very run against it, ``exact_fp32`` (the same step on the exact
fp32-MFMA path, 2 steps), and ``rccl`` for N>1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (the 5 PF headline is 2:1 sparse)
SPLIT_PASSES = 3                   # fp32 product = a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, three f16 MFMAs, fp32 accumulate
PUBLISHED_TRAIN_ITERS_PER_S = 94.0  # 1.7 s / 32-image T=5 training step on 4 unknown GPUs (log.md:3)
DOMINANT = ('conv_tile_fwd', 'conv_tile_dgrad', 'conv_tile_wgrad')
CATS = DOMINANT + ('dec_out', 'dec_out_dgrad', 'dec_out_wgrad', 'dec_out_bwd', 'dec_l0', 'l0_reduce', 'l0_slot_sum', 'pixel_pass1',
                   'pixel_pass2', 'refine_conv', 'refine_head', 'refine_wgrad', 'refine_dgrad', 'refine_bias_grad', 'head_bwd')


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
                    help='3x3 convs: 1 = fp32 operands split into fp16 hi+lo, 3 f16 MFMAs (default); 0 = exact fp32 MFMA')
    ap.add_argument('--graph', type=int, choices=[0, 1], default=0,
                    help='1 = replay the step through hipGraphs (library option "graph"); event brackets are then taken '
                         'from an extra untimed pass')
    ap.add_argument('--no-adam', action='store_true', help='time forward + backward only')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-exact-fp32', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=None)
    return ap.parse_args()


def build_model(args, device):
    import torch
    from iodine_amd import IODINE, synth
    from iodine_amd.model import clevr6_arch, dsprites_arch
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
    """Time the CPU oracle (PyTorch-CPU restatement of the reference) on a bounded sample of the same workload: all host
    cores (<= 64 threads) at batch 4 (CLEVR) / 8 (dSprites), plus one repetition at 8 threads for comparability with the
    survey container.  Returns the JSON object and what the parity check needs."""
    import torch
    from iodine_amd import synth
    from oracle import iodine_oracle as O
    oa = O.Arch(dim_latent=arch.DIM_LATENT, iters=arch.ITERS, slots=arch.SLOTS, sigma=arch.SIGMA,
                img_size=arch.IMG_SIZE, ref_chan=arch.REF.CONV_CHAN, ref_layers=arch.REF.CONV_LAYERS,
                ref_mlp=arch.REF.MLP_UNITS, dec_chan=arch.DEC.CONV_CHAN, dec_layers=arch.DEC.CONV_LAYERS)
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    Bc = args.cpu_batch or (4 if args.config == 'clevr6' else 8)
    p = {k: torch.from_numpy(v) for k, v in params.items()}
    x = torch.from_numpy(synth.make_images(Bc, oa.img_size, seed=0))
    eps = torch.from_numpy(synth.make_eps(oa.iters, Bc, oa.slots, oa.dim_latent, seed=1))
    fn = (lambda: O.reconstruct(x, eps, p, oa)) if mode == 'infer' else (lambda: O.train_step_grads(x, eps, p, oa))
    torch.set_num_threads(threads)
    fn()                                                    # warm-up
    reps, t0 = 0, time.perf_counter()
    while reps < 2 or (time.perf_counter() - t0 < 10.0 and reps < 8):
        out = fn()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    dt8 = None
    if threads > 8:
        torch.set_num_threads(8)
        t1 = time.perf_counter()
        fn()
        dt8 = time.perf_counter() - t1
        torch.set_num_threads(threads)
    ref_elbos = (out['elbos'] if mode == 'infer' else out[0]['elbos']).detach().double().numpy()
    ref_grads = None if mode == 'infer' else out[1]
    cb = dict(value=round(Bc * oa.iters / dt, 4), unit='image-refinement-iters/s', cores=threads, kind='port',
              sample=f'{mode} step, batch {Bc} of the same workload (weights, images, eps), {reps} reps after 1 warm-up, '
                     f'{dt * 1e3:.0f} ms/step; oracle/iodine_oracle.py (PyTorch-CPU fp32, the ATen arithmetic the reference runs)',
              ms_per_step=round(dt * 1e3, 1))
    if dt8 is not None:
        cb['threads8'] = dict(value=round(Bc * oa.iters / dt8, 4), ms_per_step=round(dt8 * 1e3, 1), cores=8,
                              sample='same sample, 1 repetition at torch.set_num_threads(8)')
    return cb, (x, eps, ref_elbos, ref_grads)


PMC_PIPE = None      # matrix-pipe utilisation and shader clock of the same PMC passes (the chip clocks to its power budget)


def pmc_traffic(args, B, K):
    """HBM bytes per launch of the dominant kernels from the tracked PMC file (tools/pmc_to_json.py writes it from rocprofv3
    --pmc passes: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE).  Quoted only while the file describes THIS tree (source
    digest) and THIS shape; otherwise null."""
    path = os.path.join(ROOT, 'profiles', 'r02_pmc.json')
    if not os.path.exists(path):
        return None, 'profiles/r02_pmc.json missing'
    try:
        from iodine_amd.build import source_digest
        rec = json.load(open(path))
        if rec.get('csrc_sha256') != source_digest():
            return None, 'profiles/r02_pmc.json was measured on other kernel sources (digest mismatch)'
        shape = rec.get('shape', {})
        if (shape.get('config'), shape.get('batch'), shape.get('slots')) != (args.config, B, K) or args.conv_precision != 1:
            return None, 'profiles/r02_pmc.json was measured on another shape'
        per = {k: v['hbm_bytes_per_launch'] for k, v in rec['kernels'].items()}
        global PMC_PIPE
        PMC_PIPE = {k: {f: v[f] for f in ('mfma_util', 'clock_ghz', 'mfma_rate_of_2p4ghz_peak') if f in v}
                    for k, v in rec['kernels'].items()}
        return per, f"profiles/r02_pmc.json ({rec.get('commit', '?')[:10]})"
    except Exception as e:                                  # a malformed file must not break the bench line
        return None, f'profiles/r02_pmc.json unreadable: {e}'


def main():
    args = parse()
    from iodine_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        # started the way the single-GPU run is started: become the launcher of N ranks (one process per GPU)
        sys.exit(launch.spawn(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    from iodine_amd import parallel, synth

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if rank == 0:
            print(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); start it as '
                  f'`python bench.py --gpus {args.gpus}` or under torch.distributed.run --nproc-per-node {args.gpus}',
                  file=sys.stderr, flush=True)
        sys.exit(2)
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < max(world, 1) or local >= ndev:
        if rank == 0:
            print(f'bench.py: --gpus {world} needs {world} visible ROCm devices on this node, found {ndev} '
                  f'(rank {rank} of {world}, LOCAL_RANK {local})', file=sys.stderr, flush=True)
        sys.exit(3)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)                 # before the process group: RCCL binds the communicator to this device
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    model, arch, params = build_model(args, device)
    B, T, K, S = args.batch, arch.ITERS, arch.SLOTS, arch.IMG_SIZE
    x = torch.from_numpy(synth.make_images(B, S, seed=0, first_index=rank * B)).to(device)
    model.manual_seed(1234 + rank)                # every rank draws its own noise (library Philox, inside the step)

    opt = None
    if args.mode == 'infer':
        def step():
            return model.reconstruct(x)
    else:
        from iodine_amd.optim import make_optimizer
        opt = None if args.no_adam else make_optimizer(model, base_lr=3e-4, weight_decay=0.0)   # configs/clevr6_prop.yaml:19-20

        def step():
            loss = model(x)
            model.zero_grad(set_to_none=True)                             # train.py:62 (the flat gradient buffer is replaced)
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

    def max_over_ranks(v):
        if world == 1:
            return v
        t = torch.tensor([v], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    model.set_option('conv_precision', args.conv_precision)
    model.set_option('graph', args.graph)
    for _ in range(args.warmup):
        step()
    model.set_option('profile', 0 if args.graph else 1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    model.set_option('profile', 0)
    ms_per_step = dt / args.steps * 1e3
    value = world * B * T / (dt / args.steps)

    def read_prof():
        out = {}
        for cat in CATS:
            tot, cnt = model.profile_read(cat)
            if cnt:
                out[cat] = dict(ms_total=round(tot, 3), launches=cnt, ms_avg=round(tot / cnt, 4))
        return out

    prof = read_prof()                                   # timed region: conv_tile_* only (graph mode: nothing)
    timed_events = bool(prof)

    # secondary measurement (not `value`): the other step type on the same workload
    other = None
    if args.mode == 'train':
        def istep():
            return model.reconstruct(x)
        istep()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            istep()
        barrier()
        dti = max_over_ranks((time.perf_counter() - t1) / args.steps)
        other = dict(step='infer (reconstruct: T iterations + final decode)', ms_per_step=round(dti * 1e3, 3),
                     image_refinement_iters_per_s=round(world * B * T / dti, 2))

    # ---- roofline of the dominant kernel: the decoder 3x3 conv C->C (fwd + dgrad + wgrad launches) ----
    # Inside the timed region only the dominant launches were bracketed with events (profile level 1: every category
    # costs ~1 ms per step in event records); the per-category table comes from two extra, untimed steps at level 2.
    C_ = arch.DEC.CONV_CHAN
    flops_per_launch = 2.0 * C_ * C_ * 9 * S * S * B * K
    model.set_option('graph', 0)
    model.set_option('profile', 2)
    for _ in range(2):
        step()
    barrier()
    model.set_option('profile', 0)
    prof_all = read_prof()
    for cat, v in prof_all.items():
        prof.setdefault(cat, dict(v, note='untimed pass'))
    dom_ms = sum(prof[c]['ms_total'] for c in DOMINANT if c in prof)
    dom_n = sum(prof[c]['launches'] for c in DOMINANT if c in prof)
    achieved = flops_per_launch / (dom_ms / dom_n * 1e-3) / 1e12 if dom_n else 0.0
    if args.conv_precision == 1:
        # the kernel executes SPLIT_PASSES f16 MFMAs per algorithmic (fp32-class) multiply-add: the peak for ALGORITHMIC
        # FLOPs is the dense f16 MFMA peak divided by the number of passes
        peak = PEAK_F16_MFMA_TFLOPS / SPLIT_PASSES
        kname = (f'conv3x3_ws_f16x3_kernel<{C_},EPI> + conv3x3_wgrad_f16x3_ws_kernel<{C_},{C_}> (decoder 3x3 conv {C_}->{C_}: '
                 f'fwd, dgrad, wgrad launches; fp32 in/out, operands split into f16 hi+lo, 3 f16 MFMAs, fp32 accumulate)')
    else:
        peak = PEAK_F32_MFMA_TFLOPS
        kname = (f'conv3x3_tile_kernel<{C_},{C_}> + conv3x3_wgrad_tile_kernel<{C_},{C_}> (decoder 3x3 conv {C_}->{C_}: '
                 f'fwd, dgrad, wgrad launches; exact fp32 MFMA)')
    per_kernel_traffic, traffic_src = pmc_traffic(args, B, K)
    traffic = None
    if per_kernel_traffic and dom_n:
        # launch-weighted mean over the forward / data-gradient / weight-gradient launches, like `achieved`
        w = {c: prof[c]['launches'] for c in DOMINANT if c in prof}
        if all(c in per_kernel_traffic for c in w):
            traffic = sum(per_kernel_traffic[c] * n for c, n in w.items()) / sum(w.values())
    per_form = {c: dict(ms_avg=prof[c]['ms_avg'], tflops=round(flops_per_launch / (prof[c]['ms_avg'] * 1e-3) / 1e12, 1),
                        frac=round(flops_per_launch / (prof[c]['ms_avg'] * 1e-3) / 1e12 / peak, 4))
                for c in DOMINANT if c in prof}
    roofline = dict(bound='mfma', kernel=kname, achieved=round(achieved, 2), peak=round(peak, 1), unit='TFLOP/s',
                    frac=round(achieved / peak, 4), traffic=traffic, traffic_source=traffic_src,
                    traffic_per_kernel=per_kernel_traffic, matrix_pipe_pmc=PMC_PIPE,
                    flops_per_launch=flops_per_launch, avg_launch_ms=round(dom_ms / max(dom_n, 1), 4),
                    launches=dom_n, events_in_timed_region=timed_events, per_form=per_form,
                    kernel_time_share=round(dom_ms / max(dom_n, 1) * (dom_n / (args.steps if timed_events else 2)) / ms_per_step, 4),
                    executed_mfma_tflops=round(achieved * (SPLIT_PASSES if args.conv_precision == 1 else 1), 1),
                    algorithmic_hbm_bytes_per_launch=2.0 * B * K * S * S * C_ * 4,
                    hbm_gbps_algorithmic=round(2.0 * B * K * S * S * C_ * 4 / (dom_ms / max(dom_n, 1) * 1e-3) / 1e9, 1))

    cfg_name = 'CLEVR6 128x128' if args.config == 'clevr6' else 'multi-dSprites 64x64'
    what = ('reconstruct: T iterations + final decode' if args.mode == 'infer'
            else 'forward + backward' + ('' if args.no_adam else ' + fused Adam step') + ', noise drawn inside the step')
    out = dict(metric='refinement_iters_per_s', value=round(value, 2), unit='image-refinement-iters/s',
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3),
               higher_is_better=True, scaling='weak',
               vs_baseline=(round(value / PUBLISHED_TRAIN_ITERS_PER_S, 3) if args.mode == 'train' and args.config == 'clevr6' else None),
               dtype=('f32 (3xf16-split MFMA convs, f32 accumulate)' if args.conv_precision == 1 else 'f32'), data='synthetic',
               config=dict(workload=f'{cfg_name}, K={K}, T={T}, batch {B}/GPU, {args.mode} step ({what})',
                           step=args.mode, global_batch=B * world, slots=K, iters=T, img_size=S, hip_graph=bool(args.graph),
                           parallelism=f'dp{world} (images sharded over ranks; '
                                       f'{"one RCCL all-reduce of the flat gradient buffer per step" if args.mode == "train" else "no data-path collective"})'),
               batch_iters_per_s=round(T / (dt / args.steps), 3), roofline=roofline, kernels=prof)
    if other:
        out['inference_step'] = other

    if world > 1:
        # the collective of the step, measured on its own: all-reduce of a buffer the size of the flat gradients
        nflat = sum(p.numel() for p in model.parameters())
        buf = torch.zeros(nflat, device=device)
        for _ in range(3):
            dist.all_reduce(buf)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        us = max_over_ranks(e0.elapsed_time(e1) / 20 * 1e3)
        out['rccl'] = dict(world_size=dist.get_world_size(), backend=dist.get_backend(), allreduce_bytes=nflat * 4,
                           allreduce_us=round(us, 1), per_step=1 if args.mode == 'train' else 0)

    if rank == 0 and world == 1 and args.mode == 'train' and not args.no_exact_fp32 and args.conv_precision == 1:
        # the precision trade on the record: the same step on the exact fp32-MFMA path (v_mfma_f32_32x32x2_f32)
        model.set_option('conv_precision', 0)
        step()
        model.set_option('profile', 1)
        barrier()
        t2 = time.perf_counter()
        for _ in range(2):
            step()
        barrier()
        dtx = (time.perf_counter() - t2) / 2
        model.set_option('profile', 0)
        px = read_prof()
        xm = sum(px[c]['ms_total'] for c in DOMINANT if c in px)
        xn = sum(px[c]['launches'] for c in DOMINANT if c in px)
        xa = flops_per_launch / (xm / xn * 1e-3) / 1e12 if xn else 0.0
        out['exact_fp32'] = dict(ms_per_step=round(dtx * 1e3, 3), steps=2, dominant_avg_launch_ms=round(xm / max(xn, 1), 4),
                                 achieved_tflops=round(xa, 2), frac_of_157_3=round(xa / PEAK_F32_MFMA_TFLOPS, 4),
                                 speedup_of_default_path=round(dtx * 1e3 / ms_per_step, 2))
        model.set_option('conv_precision', 1)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, (xc, ec, ref_elbos, ref_grads) = cpu_baseline(args, arch, params, args.mode)
        out['cpu_baseline'] = cb
        # parity of this very run against the oracle on the CPU sample (gates: ELBO 1e-3, gradient rel-L2 1e-3, north_star /
        # SURVEY 8d); the timed training steps moved the weights (Adam), so the initial ones - what the oracle was given -
        # are loaded back first
        import numpy as np
        with torch.no_grad():
            model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        if args.mode == 'infer':
            model.reconstruct(xc.to(device), ec.to(device))
        else:
            model.zero_grad(set_to_none=True)
            model(xc.to(device), ec.to(device)).backward()
            num = sum(float(((p.grad.double().cpu() - ref_grads[n].double()) ** 2).sum()) for n, p in model.named_parameters())
            den = sum(float((ref_grads[n].double() ** 2).sum()) for n, _ in model.named_parameters())
            out['grad_rel_l2_vs_cpu'] = float(np.sqrt(num / den))
        got = model.elbo_terms[:, 0].double().cpu().numpy()
        n = min(len(got), len(ref_elbos))
        out['elbo_rel_err_vs_cpu'] = float(abs(got[:n] - ref_elbos[:n]).max() / abs(ref_elbos[:n]).max())
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
