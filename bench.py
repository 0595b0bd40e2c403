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

The timed region is EXACTLY ``--steps`` steps (``value``, ``ms_per_step``).  When that region is shorter than 6 s the same step
keeps running afterwards until 6 s of continuous GPU work have passed (``sustained``: steps_run, ms_per_step - a second,
longer measurement of the same loop, and long enough for an external utilisation sampler to see the GPU busy).

Extra objects on the JSON line:
  * ``roofline``: the dominant kernel (split-fp16 MFMA 3x3 conv C->C, forward / data-gradient / weight-gradient launches);
    duration from HIP events recorded on the launch stream inside the timed region; ``traffic`` from the tracked PMC file
    profiles/r*_pmc.json while its source digest still matches the tree;
  * ``roofline_hbm``: the other regime (SURVEY 8d "report both") - of the HBM-bound helper kernels the one that loses the most
    time against the 8 TB/s roof (time x (1 - frac)), with the whole candidate table; algorithmic bytes per launch from the
    shapes, durations from HIP events of an extra pass that brackets every launch;
  * ``inference_step``: the reconstruct step on the same workload with its own ``roofline`` (forward + data-gradient launches);
  * ``configs``: BASELINE configs[1] (dSprites 64x64, K=6, T=5, B=32) and the per-GPU shard of configs[4] (K=11, T=7, B=8),
    5 steps each of both step types - side measurements, not part of ``value``;
  * ``cpu_baseline``: the CPU oracle timed on this box's host cores on a bounded sample (rank 0 at N=1 only), with ELBO and
    gradient parity of this very run against it; ``exact_fp32``: the same step on the exact fp32-MFMA path;
  * ``rccl`` for N>1: world size, all-reduce time, and whether the replicas held identical parameters before the first and
    after the last step.
"""
import argparse
import json
import math
import os
import sys
import time

# one process per GPU over RCCL: the host driver only supports dmabuf IPC.  Under the driver's own
# `python -m torch.distributed.run ... bench.py --gpus N` nothing else sets this (iodine_amd/launch.py does for the self-spawn
# path), and it has to be in the environment before the HIP runtime is loaded by `import torch`.
if int(os.environ.get('WORLD_SIZE', '1')) > 1 or '--gpus' in sys.argv:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (the 5 PF headline is 2:1 sparse)
PEAK_HBM_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s
SUSTAIN_S = 6.0                    # minimum length of the continuous step loop (timed region + its continuation)
SPLIT_PASSES = 3                   # fp32 product = a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, three f16 MFMAs, fp32 accumulate
PUBLISHED_TRAIN_ITERS_PER_S = 94.0  # 1.7 s / 32-image T=5 training step on 4 unknown GPUs (log.md:3)
DOMINANT = ('conv_tile_fwd', 'conv_tile_dgrad', 'conv_tile_wgrad')
FP32_KNAME = ('conv3x3_ws_f16x3_kernel<{C},EPI,F32=true> + conv3x3_wgrad_f32_ws_kernel<{C}> (decoder 3x3 conv {C}->{C}: fwd, dgrad, wgrad launches; '
              'weight-stationary / persistent, exact fp32 MFMA: v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32)')
PROFILE_STRIDE = 7                 # timed region: every 7th launch of each dominant form is bracketed with HIP events (library option profile_stride)
CATS = DOMINANT + ('dec_out', 'dec_out_dgrad', 'dec_out_wgrad', 'dec_out_bwd', 'dec_l0', 'l0_reduce', 'l0_slot_sum', 'pixel_pass1',
                   'pixel_pass2', 'refine_conv', 'refine_l0', 'refine_l0f', 'refine_head', 'refine_wgrad', 'refine_dgrad', 'refine_bwd01', 'refine_bias_grad', 'head_bwd')


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
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the cfg2 / cfg5-shard side measurements')
    ap.add_argument('--no-sustain', action='store_true', help='do not continue a short timed region to 6 s of GPU work')
    ap.add_argument('--cpu-batch', type=int, default=None)
    ap.add_argument('--option', action='append', default=[], metavar='KEY=VALUE',
                    help='library option for the timed model (A/B of kernel selections, e.g. refine_l0_fused=0); recorded in config.options')
    return ap.parse_args()


def build_model(config, slots, iters, device):
    import torch
    from iodine_amd import IODINE, synth
    from iodine_amd.model import clevr6_arch, dsprites_arch
    if config == 'clevr6':
        arch = clevr6_arch(slots=slots or 7, iters=iters or 5)
    else:
        arch = dsprites_arch(slots=slots or 6, iters=iters or 5)
    model = IODINE(arch)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    params = synth.make_params(shapes, seed=0)              # torch-default-init bounds, deterministic bytes
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return model.to(device), arch, params


def cpu_baseline(args, arch, params, mode):
    """Time the CPU oracle (PyTorch-CPU restatement of the reference) on a bounded sample of the same workload, SURVEY 8d: for
    torch.set_num_threads in {8, 16, 32, min(all host cores, 64)} one batch-1 warm-up + 2 timed repetitions of ONE step at the same batch
    (2 images CLEVR / 8 dSprites); `value` is the BEST thread count's mean, the others stay beside it (`sweep`).  A thread count whose
    first repetition is already > 2.5x slower than the best so far is not repeated (bounds the CPU time to ~30 s).  Returns the JSON
    object and what the parity check needs."""
    import torch
    from iodine_amd import synth
    from oracle import iodine_oracle as O
    oa = O.Arch(dim_latent=arch.DIM_LATENT, iters=arch.ITERS, slots=arch.SLOTS, sigma=arch.SIGMA,
                img_size=arch.IMG_SIZE, ref_chan=arch.REF.CONV_CHAN, ref_layers=arch.REF.CONV_LAYERS,
                ref_mlp=arch.REF.MLP_UNITS, dec_chan=arch.DEC.CONV_CHAN, dec_layers=arch.DEC.CONV_LAYERS)
    cores = os.cpu_count() or 1
    Bc = args.cpu_batch or (2 if args.config == 'clevr6' else 8)
    p = {k: torch.from_numpy(v) for k, v in params.items()}
    x = torch.from_numpy(synth.make_images(Bc, oa.img_size, seed=0))
    eps = torch.from_numpy(synth.make_eps(oa.iters, Bc, oa.slots, oa.dim_latent, seed=1))

    def fn(n):
        xe = (x[:n], eps[:, :n].contiguous())
        return O.reconstruct(*xe, p, oa) if mode == 'infer' else O.train_step_grads(*xe, p, oa)

    sweep, best, out = {}, None, None
    # (all host cores, capped at 64: on the 256-core build boxes a 256-thread step of this sample took 223 s in round 5 - oversubscribed)
    for threads in sorted({t for t in (8, 16, 32, min(cores, 64)) if t <= cores} or {cores}):
        torch.set_num_threads(threads)
        fn(1)                                               # warm-up (thread pool, oneDNN primitive cache)
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            o = fn(Bc)
            times.append(time.perf_counter() - t0)
            if out is None:
                out = o                                     # (the arithmetic does not depend on the thread count at the parity gates)
            if best is not None and times[0] > 2.5 * best[1]:
                break
        mean = sum(times) / len(times)
        sweep[str(threads)] = dict(cores=threads, reps=len(times), ms_per_step=round(mean * 1e3, 1), ms_min=round(min(times) * 1e3, 1),
                                   value=round(Bc * oa.iters / mean, 4))
        if len(times) == 2 and (best is None or mean < best[1]):
            best = (threads, mean, len(times))
    threads, dt, reps = best
    torch.set_num_threads(min(cores, 64))
    ref_elbos = (out['elbos'] if mode == 'infer' else out[0]['elbos']).detach().double().numpy()
    ref_grads = None if mode == 'infer' else out[1]
    cb = dict(value=round(Bc * oa.iters / dt, 4), unit='image-refinement-iters/s', cores=threads, kind='port', reps=reps, host_cores=cores,
              sample=f'{mode} step, batch {Bc} of the same workload (weights, images, eps): batch-1 warm-up + {reps} timed steps per thread count, '
                     f'best of torch.set_num_threads in {sorted(int(k) for k in sweep)} = {threads} threads, mean {dt * 1e3:.0f} ms/step; '
                     f'oracle/iodine_oracle.py (PyTorch-CPU fp32, the ATen arithmetic the reference runs)',
              ms_per_step=round(dt * 1e3, 1), sweep=sweep)
    return cb, (x, eps, ref_elbos, ref_grads)


def pmc_file(strict=False, config='clevr6'):
    """The newest tracked PMC record (profiles/rNN_pmc.json; strict: rNN_pmc_strict.json = the same passes over --conv-precision 0;
    config dsprites: rNN_pmc_dsprites.json = the passes over --config dsprites, BASELINE configs[1])."""
    import glob
    pat = 'r[0-9][0-9]_pmc_dsprites.json' if config == 'dsprites' else 'r[0-9][0-9]_pmc_strict.json' if strict else 'r[0-9][0-9]_pmc.json'
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pat)))
    return files[-1] if files else None


def pmc_record(args, B, K, strict=None, config=None):
    """HBM bytes per launch (every profiled category) and matrix-pipe utilisation / shader clock of the dominant kernels from the
    tracked PMC file (tools/pmc_to_json.py writes it from rocprofv3 --pmc passes: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE).
    Quoted only while the file describes THIS tree (source digest) and THIS shape; otherwise (None, None, reason)."""
    strict = (args.conv_precision == 0) if strict is None else strict
    config = config or args.config
    path = pmc_file(strict, config)
    if not path:
        return None, None, 'no profiles/rNN_pmc_strict.json' if strict else 'no profiles/rNN_pmc.json'
    name = os.path.relpath(path, ROOT)
    try:
        from iodine_amd.build import source_digest
        rec = json.load(open(path))
        if rec.get('csrc_sha256') != source_digest():
            return None, None, f'{name} was measured on other kernel sources (digest mismatch)'
        shape = rec.get('shape', {})
        if (shape.get('config'), shape.get('batch'), shape.get('slots')) != (config, B, K) or bool(rec.get('conv_precision', 1) == 0) != bool(strict):
            return None, None, f'{name} was measured on another shape'
        per = {k: v['hbm_bytes_per_launch'] for k, v in rec['kernels'].items() if 'hbm_bytes_per_launch' in v}
        pipe = {k: {f: v[f] for f in ('mfma_util', 'clock_ghz', 'mfma_rate_of_2p4ghz_peak') if f in v}
                for k, v in rec['kernels'].items() if 'mfma_util' in v}
        return per, pipe, f"{name} ({rec.get('commit', '?')[:10]})"
    except Exception as e:                                  # a malformed file must not break the bench line
        return None, None, f'{name} unreadable: {e}'


def hbm_algorithmic_bytes(arch, B, mode, ran=None):
    """Algorithmic HBM bytes PER LAUNCH (mean over the launches of one category) of the HBM-bound helper kernels:
    every input tensor read once, every output written once, fp32; weights and KB-sized side buffers not counted."""
    K, T, S = arch.SLOTS, arch.ITERS, arch.IMG_SIZE
    Cd, Cr, Dr = arch.DEC.CONV_CHAN, arch.REF.CONV_CHAN, arch.REF.CONV_LAYERS
    N, P = B * K, S * S
    act = 4.0 * N * P * Cd                                   # one decoder activation / gradient tensor
    out4 = 16.0 * N * P                                      # decoder output {rgb, mask logit} resp. its gradient
    x4 = 16.0 * B * P
    enc = 48.0 * N * P + 32.0 * B * P                        # split refinement input: 12 floats per slot-pixel + 8 per image-pixel
    b = {
        'dec_l0': act,                                       # broadcast layer: write only (4 MB class map read)
        'dec_out': act + out4,
        'dec_out_bwd': 2 * act + out4,                       # read activation + output gradient, write data gradient
        'dec_out_dgrad': 2 * act + out4,
        'dec_out_wgrad': act + out4,
        'pixel_pass1': 2 * out4 + x4,
        'pixel_pass2': out4 + x4 + enc,
        'l0_reduce': 4.0 * N * S * (S // 16) * (4 if mode == 'train' else 3) * Cd,
    }
    # refinement stack, stride 2: layer l reads [N][s][s][C_in], writes [N][s/2][s/2][Cr]
    per_layer, s = [], S
    for l in range(Dr):
        per_layer.append((enc if l == 0 else 4.0 * N * s * s * Cr, 4.0 * N * (s // 2) * (s // 2) * Cr))
        s //= 2
    # first layer: per-image part (reads 8 floats per image-pixel, writes a map) + per-slot part (reads 12 floats per
    # slot-pixel and the map, writes the activation): two launches
    b['refine_l0'] = (per_layer[0][0] + per_layer[0][1] + 2 * 4.0 * B * (S // 2) * (S // 2) * Cr) / 2.0
    # round 4: encoding + first layer in one kernel (refine_l0f): reads the decoder output and the image, writes the layer's output; the
    # encoding itself is only written in training (the backward reads it)
    b['refine_l0f'] = out4 + x4 + per_layer[0][1] + (enc if mode == 'train' else 0.0)
    if Dr > 1:
        b['refine_conv'] = sum(i + o for i, o in per_layer[1:]) / (Dr - 1)
    if mode == 'train':                                      # one batch of T * N slot-images per layer
        # round 4: the data gradient of layer 1 and the weight gradient of layer 0 are one launch (refine_bwd01: reads d(out 1), the
        # saved activation 0 and the encoding; d(pre-activation 0) is not stored) - the other layers as before
        # (which form ran is read off the categories that actually launched - the library's own predicates (power-of-two size, option
        # values) decide, not a copy of them here)
        fused01 = ('refine_bwd01' in ran) if ran is not None else (Dr > 1 and Cr == 64 and S >= 64 and (S & (S - 1)) == 0)
        if fused01:
            b['refine_bwd01'] = T * (per_layer[0][0] + per_layer[0][1] + per_layer[1][1])
            b['refine_wgrad'] = T * sum(i + o for i, o in per_layer[1:]) / (Dr - 1)
            if Dr > 2:
                b['refine_dgrad'] = T * sum(2 * i + o for i, o in per_layer[2:]) / (Dr - 2)
        else:
            b['refine_wgrad'] = T * sum(i + o for i, o in per_layer) / Dr
            if Dr > 1:
                b['refine_dgrad'] = T * sum(2 * i + o for i, o in per_layer[1:]) / (Dr - 1)   # read d(out) + saved act, write d(in)
    return b


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
    def fail_line(code, msg, **extra):
        """Every failure path of the N > 1 start-up leaves a diagnosable record on rank 0: the message on stderr AND a JSON
        object with the rccl block (world size, backend, rendezvous) on stdout, so that a hang or a refusal can be read off the tail."""
        if rank == 0:
            print(f'bench.py: {msg}', file=sys.stderr, flush=True)
            print(json.dumps(dict(error=msg, exit_code=code, n_gpus=world,
                                  rccl=dict(world_size=world, rank=rank, local_rank=local,
                                            backend=('gloo' if os.environ.get('IODINE_BENCH_SHARE_DEVICE') == '1' else 'nccl'),
                                            master=f"{os.environ.get('MASTER_ADDR', '127.0.0.1')}:{os.environ.get('MASTER_PORT', '?')}",
                                            hsa_ipc_mode_legacy=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), **extra))), flush=True)
        sys.exit(code)

    if args.gpus != world:
        fail_line(2, f'--gpus {args.gpus} but the launcher started {world} rank(s); start it as '
                     f'`python bench.py --gpus {args.gpus}` or under torch.distributed.run --nproc-per-node {args.gpus}')
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    # test hook (tests/test_gpu_multirank.py on a 1-GPU box): IODINE_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and runs the
    # collectives over gloo - RCCL refuses two ranks on one device.  Everything else of the N > 1 path is the code that ships.
    share = os.environ.get('IODINE_BENCH_SHARE_DEVICE') == '1' and world > 1
    if share:
        local = 0
    if ndev < (1 if share else max(world, 1)) or local >= ndev:
        fail_line(3, f'--gpus {world} needs {world} visible ROCm devices on this node, found {ndev} '
                     f'(rank {rank} of {world}, LOCAL_RANK {local})', visible_devices=ndev)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)                 # before the process group: RCCL binds the communicator to this device
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import datetime
        try:
            if share:
                dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device,
                                        timeout=datetime.timedelta(seconds=300))
                # the first collective builds the RCCL communicator (xGMI rings): do it here, where a failure is attributable
                t0c = torch.zeros(1, device=device)
                dist.all_reduce(t0c)
                torch.cuda.synchronize()
        except Exception as e:                      # noqa: BLE001 - whatever RCCL / the rendezvous raised goes on the record
            fail_line(4, f'process group start-up failed on rank {rank}: {type(e).__name__}: {e}', visible_devices=ndev)

    model, arch, params = build_model(args.config, args.slots, args.iters, device)
    B, T, K, S = args.batch, arch.ITERS, arch.SLOTS, arch.IMG_SIZE
    x = torch.from_numpy(synth.make_images(B, S, seed=0, first_index=rank * B)).to(device)
    model.manual_seed(1234 + rank)                # every rank draws its own noise (library Philox, inside the step)
    # replicas start from rank 0's parameters (DataParallel re-broadcasts them on every forward, lib/modeling/build.py:11-12)
    # and are CHECKED to be bitwise identical before the first step and after the last one
    parallel.broadcast_parameters(model, 0)
    replicas_before = parallel.replicas_identical(model.parameters())

    def make_step(m, xb, mode, adam=True):
        if mode == 'infer':
            return (lambda: m.reconstruct(xb)), None
        from iodine_amd.optim import make_optimizer
        opt = make_optimizer(m, base_lr=3e-4, weight_decay=0.0) if adam else None   # configs/clevr6_prop.yaml:19-20

        def step():
            loss = m(xb)
            m.zero_grad(set_to_none=True)                                 # train.py:62 (the flat gradient buffer is replaced)
            loss.backward()
            parallel.allreduce_gradients(m.parameters(), world)           # one RCCL all-reduce of the flat grads
            if opt is not None:
                opt.step()
            return loss
        return step, opt

    step, opt = make_step(model, x, args.mode, not args.no_adam)

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

    local_dt = [0.0]                                     # this rank's own clock over the last timed() region

    def timed(fn, n):
        """n calls of fn bracketed by barrier + synchronize on both sides; seconds, max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        local_dt[0] = time.perf_counter() - t0
        return max_over_ranks(local_dt[0])

    def read_prof(m=None):
        out = {}
        for cat in CATS:
            tot, cnt = (m or model).profile_read(cat)
            _, seen = (m or model).profile_read('seen:' + cat)              # every launch of the category, bracketed or not
            if cnt:
                out[cat] = dict(ms_total=round(tot, 3), launches=cnt, ms_avg=round(tot / cnt, 4), **({'launches_seen': seen} if seen else {}))
        return out

    model.set_option('conv_precision', args.conv_precision)
    model.set_option('graph', args.graph)
    for kv in args.option:
        k, v = kv.split('=')
        model.set_option(k, float(v))
    for _ in range(args.warmup):
        step()
    model.set_option('profile_stride', PROFILE_STRIDE)
    model.set_option('profile', 0 if args.graph else 1)
    dt = timed(step, args.steps)                                          # ---- THE timed region: exactly --steps steps ----
    ms_per_step = dt / args.steps * 1e3
    rank_ms = [local_dt[0] / args.steps * 1e3]
    if world > 1:                                        # every rank's own ms/step of the timed region (jitter between ranks)
        tl = torch.tensor([rank_ms[0]], device=device, dtype=torch.float64)
        tg = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(tg, tl)
        rank_ms = [float(t.item()) for t in tg]
    value = world * B * T / (dt / args.steps)
    prof = read_prof()                                   # timed region: conv_tile_* only (graph mode: nothing)
    timed_events = bool(prof)
    model.set_option('profile', 0)

    # the same loop continued: at least SUSTAIN_S seconds of back-to-back steps in all (its own, longer measurement)
    sustained = None
    if not args.no_sustain and dt < SUSTAIN_S:
        n_more = int(math.ceil((SUSTAIN_S - dt) / (dt / args.steps)))
        dts = timed(step, n_more)
        sustained = dict(steps_run=args.steps + n_more, seconds=round(dt + dts, 3), continuation_steps=n_more,
                         continuation_ms_per_step=round(dts / n_more * 1e3, 3),
                         note=f'value / ms_per_step are the first {args.steps} steps (the requested timed region); the loop then '
                              f'continued for {n_more} more steps so that the GPU is busy for >= {SUSTAIN_S:.0f} s in one stretch')

    # ---- roofline of the dominant kernel: the decoder 3x3 conv C->C (fwd + dgrad + wgrad launches) ----
    # Inside the timed region only the dominant launches were bracketed with events (profile level 1: every category
    # costs ~1 ms per step in event records); the per-category table comes from extra, untimed steps at level 2.
    C_ = arch.DEC.CONV_CHAN
    flops_per_launch = 2.0 * C_ * C_ * 9 * S * S * B * K
    peak = PEAK_F16_MFMA_TFLOPS / SPLIT_PASSES if args.conv_precision == 1 else PEAK_F32_MFMA_TFLOPS
    per_kernel_traffic, pmc_pipe, traffic_src = pmc_record(args, B, K)

    def mfma_roofline(pr, steps_counted, step_ms, in_timed):
        dom_ms = sum(pr[c]['ms_total'] for c in DOMINANT if c in pr)
        dom_n = sum(pr[c]['launches'] for c in DOMINANT if c in pr)
        dom_all = sum(pr[c].get('launches_seen', pr[c]['launches']) for c in DOMINANT if c in pr)      # bracketed: every PROFILE_STRIDE-th
        achieved = flops_per_launch / (dom_ms / dom_n * 1e-3) / 1e12 if dom_n else 0.0
        if args.conv_precision == 1:
            # the kernel executes SPLIT_PASSES f16 MFMAs per algorithmic (fp32-class) multiply-add: the peak for ALGORITHMIC
            # FLOPs is the dense f16 MFMA peak divided by the number of passes
            kname = (f'conv3x3_ws_f16x3_kernel<{C_},EPI> + conv3x3_wgrad_f16x3_ws_kernel<{C_},{C_}> (decoder 3x3 conv {C_}->{C_}: '
                     f'fwd, dgrad, wgrad launches; fp32 in/out, operands split into f16 hi+lo, 3 f16 MFMAs, fp32 accumulate)')
        else:
            kname = FP32_KNAME.format(C=C_)
        traffic = None
        if per_kernel_traffic and dom_n:
            # launch-weighted mean over the forward / data-gradient / weight-gradient launches, like `achieved`
            w = {c: pr[c].get('launches_seen', pr[c]['launches']) for c in DOMINANT if c in pr}
            if all(c in per_kernel_traffic for c in w):
                traffic = sum(per_kernel_traffic[c] * n for c, n in w.items()) / sum(w.values())
        per_form = {c: dict(ms_avg=pr[c]['ms_avg'], tflops=round(flops_per_launch / (pr[c]['ms_avg'] * 1e-3) / 1e12, 1),
                            frac=round(flops_per_launch / (pr[c]['ms_avg'] * 1e-3) / 1e12 / peak, 4))
                    for c in DOMINANT if c in pr}
        avg_ms = dom_ms / max(dom_n, 1)
        return dict(bound='mfma', kernel=kname, achieved=round(achieved, 2), peak=round(peak, 1), unit='TFLOP/s',
                    frac=round(achieved / peak, 4), traffic=traffic, traffic_source=traffic_src,
                    traffic_per_kernel=({c: per_kernel_traffic[c] for c in DOMINANT if c in per_kernel_traffic}
                                        if per_kernel_traffic else None),
                    matrix_pipe_pmc=pmc_pipe, flops_per_launch=flops_per_launch, avg_launch_ms=round(avg_ms, 4),
                    launches=dom_all, launches_timed=dom_n, events_in_timed_region=in_timed,
                    event_sampling=(f'every {PROFILE_STRIDE}th launch of each form is bracketed (a pair of event records idles the GPU for ~12 us: '
                                    f'bracketing all {dom_all // max(steps_counted, 1)} launches of a step cost 0.8 ms of the step, tools/step_timeline.py); '
                                    f'{PROFILE_STRIDE} is coprime to the layer count, every layer is sampled' if in_timed and dom_all != dom_n else None),
                    per_form=per_form,
                    kernel_time_share=round(avg_ms * (dom_all / max(steps_counted, 1)) / step_ms, 4),
                    executed_mfma_tflops=round(achieved * (SPLIT_PASSES if args.conv_precision == 1 else 1), 1),
                    algorithmic_hbm_bytes_per_launch=2.0 * B * K * S * S * C_ * 4,
                    hbm_gbps_algorithmic=round(2.0 * B * K * S * S * C_ * 4 / (avg_ms * 1e-3) / 1e9, 1) if dom_n else None)

    cfg_of = {}                                              # id(arch) -> bench config name of a side model ('dsprites' has its own PMC record)

    def side_roofline(m, a, bsz, step_fn, step_ms, mode):
        """`roofline` block of a side configuration (cfg2, cfg5 shard): the dominant decoder 3x3 conv C -> C of THAT shape, launches bracketed
        with HIP events (every PROFILE_STRIDE-th) in an extra pass of 3 steps behind the timed one - the brackets cost ~12 us of idle GPU
        each, 2 % of a 6.6 ms cfg2 step, so they stay out of its timed region.  Both roofs are priced: matrix pipe (algorithmic FLOPs vs the
        dense f16 peak / 3 split passes) and HBM (every operand tensor once per launch vs 8 TB/s); `bound` names the roof the kernel is closer to."""
        Cc, Ss, Kk = a.DEC.CONV_CHAN, a.IMG_SIZE, a.SLOTS
        m.set_option('profile_stride', PROFILE_STRIDE)
        m.set_option('profile', 1)
        n = 3
        for _ in range(n):
            step_fn()
        barrier()
        m.set_option('profile', 0)
        pr = read_prof(m)
        dom_ms = sum(pr[c]['ms_total'] for c in DOMINANT if c in pr)
        dom_n = sum(pr[c]['launches'] for c in DOMINANT if c in pr)
        dom_all = sum(pr[c].get('launches_seen', pr[c]['launches']) for c in DOMINANT if c in pr)
        if not dom_n:
            return None
        fl = 2.0 * Cc * Cc * 9 * Ss * Ss * bsz * Kk
        act_bytes = 4.0 * bsz * Kk * Ss * Ss * Cc
        tensors = dict(conv_tile_fwd=2, conv_tile_dgrad=3, conv_tile_wgrad=2)     # in + out; d(out) + ELU' operand + d(in); activation + gradient
        avg = dom_ms / dom_n
        tf = fl / (avg * 1e-3) / 1e12
        pk = PEAK_F16_MFMA_TFLOPS / SPLIT_PASSES
        w = {c: pr[c].get('launches_seen', pr[c]['launches']) for c in DOMINANT if c in pr}
        by = sum(tensors[c] * act_bytes * k for c, k in w.items()) / sum(w.values())
        gbps = by / (avg * 1e-3) / 1e9
        per_form = {c: dict(ms_avg=pr[c]['ms_avg'], tflops=round(fl / (pr[c]['ms_avg'] * 1e-3) / 1e12, 1),
                            frac_mfma=round(fl / (pr[c]['ms_avg'] * 1e-3) / 1e12 / pk, 4),
                            hbm_gbps_algorithmic=round(tensors[c] * act_bytes / (pr[c]['ms_avg'] * 1e-3) / 1e9, 1),
                            frac_hbm=round(tensors[c] * act_bytes / (pr[c]['ms_avg'] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4))
                    for c in DOMINANT if c in pr}
        f_m, f_h = tf / pk, gbps / PEAK_HBM_GBPS
        hb = f_h > f_m
        # HBM bytes per launch from the PMC record of THIS shape (cfg2: profiles/rNN_pmc_dsprites.json), launch-weighted like `achieved`
        s_per, s_pipe, s_src = pmc_record(args, bsz, Kk, strict=False, config=cfg_of.get(id(a), 'no-record'))
        s_traffic = None
        if s_per and all(c in s_per for c in w):
            s_traffic = sum(s_per[c] * k for c, k in w.items()) / sum(w.values())
        return dict(bound='hbm' if hb else 'mfma',
                    kernel=f'conv3x3_ws_f16x3_kernel<{Cc},EPI>' + (f' + conv3x3_wgrad_f16x3_ws_kernel<{Cc},{Cc}>' if mode == 'train' else '')
                           + f' (decoder 3x3 conv {Cc}->{Cc}: ' + ('fwd, dgrad, wgrad' if mode == 'train' else 'fwd, dgrad') + ' launches)',
                    achieved=round(gbps if hb else tf, 2), peak=PEAK_HBM_GBPS if hb else round(pk, 1), unit='GB/s' if hb else 'TFLOP/s',
                    frac=round(max(f_m, f_h), 4), frac_mfma=round(f_m, 4), frac_hbm=round(f_h, 4), traffic=s_traffic,
                    traffic_source=s_src, traffic_per_kernel=({c: s_per[c] for c in w if c in s_per} if s_per else None),
                    matrix_pipe_pmc=({c: v for c, v in s_pipe.items() if c in w} if s_pipe else None), flops_per_launch=fl, algorithmic_hbm_bytes_per_launch=by,
                    avg_launch_ms=round(avg, 4), launches_per_step=round(dom_all / n, 1), launches_timed=dom_n, events_in_timed_region=False,
                    kernel_time_share=round(avg * (dom_all / n) / step_ms, 4), per_form=per_form)

    def hbm_roofline(pr, mode, steps_counted):
        """The HBM regime: every helper category with known algorithmic bytes -> achieved GB/s vs 8 TB/s; the reported kernel is
        the one with the largest time x (1 - frac) per step (what a perfect streaming kernel would give back)."""
        alg = hbm_algorithmic_bytes(arch, B, mode, ran=set(pr))
        cand = {}
        for c, nbytes in alg.items():
            if c not in pr:
                continue
            ms = pr[c]['ms_avg']
            gbps = nbytes / (ms * 1e-3) / 1e9
            cand[c] = dict(ms_avg=ms, launches_per_step=round(pr[c]['launches'] / steps_counted, 2), algorithmic_bytes=round(nbytes),
                           gbps=round(gbps, 1), frac=round(gbps / PEAK_HBM_GBPS, 4),
                           ms_per_step=round(pr[c]['ms_total'] / steps_counted, 3),
                           ms_per_step_above_roof=round(pr[c]['ms_total'] / steps_counted * (1.0 - gbps / PEAK_HBM_GBPS), 3),
                           traffic=(per_kernel_traffic or {}).get(c))
        if not cand:
            return None
        worst = max(cand, key=lambda c: cand[c]['ms_per_step_above_roof'])
        w = cand[worst]
        return dict(bound='hbm', kernel=worst, achieved=w['gbps'], peak=PEAK_HBM_GBPS, unit='GB/s', frac=w['frac'],
                    traffic=w['traffic'], traffic_source=traffic_src, algorithmic_bytes_per_launch=w['algorithmic_bytes'],
                    avg_launch_ms=w['ms_avg'], launches_per_step=w['launches_per_step'], events_in_timed_region=False,
                    selection='largest time x (1 - frac) per step over the HBM-bound helper categories (extra pass of '
                              f'{steps_counted} steps, every launch bracketed with HIP events)',
                    candidates=cand)

    def level2_pass(fn, n=3):
        model.set_option('graph', 0)
        model.set_option('profile', 2)
        for _ in range(n):
            fn()
        barrier()
        model.set_option('profile', 0)
        return read_prof(), n

    prof_all, n2 = level2_pass(step)
    for cat, v in prof_all.items():
        prof.setdefault(cat, dict(v, note='untimed pass'))
    roofline = mfma_roofline(prof if timed_events else prof_all, args.steps if timed_events else n2, ms_per_step, timed_events)
    roofline_hbm = hbm_roofline(prof_all, args.mode, n2)

    # secondary measurement (not `value`): the other step type on the same workload, with its own roofline
    other = None
    if args.mode == 'train':
        istep, _ = make_step(model, x, 'infer')
        istep()
        model.set_option('graph', args.graph)
        model.set_option('profile', 0 if args.graph else 1)
        dti = timed(istep, args.steps)
        iprof = read_prof()
        model.set_option('profile', 0)
        ims = dti / args.steps * 1e3
        n_inf = args.steps
        if not args.no_sustain and dti < SUSTAIN_S / 2:
            n_more = int(math.ceil((SUSTAIN_S / 2 - dti) / (dti / args.steps)))
            timed(istep, n_more)
            n_inf += n_more
        iprof_all, ni2 = level2_pass(istep)
        other = dict(step='infer (reconstruct: T iterations + final decode)', ms_per_step=round(ims, 3), steps=args.steps,
                     steps_run=n_inf, image_refinement_iters_per_s=round(world * B * T / (dti / args.steps), 2),
                     roofline=mfma_roofline(iprof if iprof else iprof_all, args.steps if iprof else ni2, ims, bool(iprof)),
                     roofline_hbm=hbm_roofline(iprof_all, 'infer', ni2), kernels=iprof_all)

    cfg_name = 'CLEVR6 128x128' if args.config == 'clevr6' else 'multi-dSprites 64x64'
    what = ('reconstruct: T iterations + final decode' if args.mode == 'infer'
            else 'forward + backward' + ('' if args.no_adam else ' + fused Adam step') + ', noise drawn inside the step')
    out = dict(metric='refinement_iters_per_s', value=round(value, 2), unit='image-refinement-iters/s',
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3),
               higher_is_better=True, scaling='weak',
               vs_baseline=(round(value / PUBLISHED_TRAIN_ITERS_PER_S, 3) if args.mode == 'train' and args.config == 'clevr6' else None),
               dtype=('f32 tensors; conv products from 3 f16 MFMAs on hi+lo splits: >= 22 bits relative to the 8x16-cell maximum (tile-relative), f32 accumulate' if args.conv_precision == 1 else 'f32'), data='synthetic',
               config=dict(workload=f'{cfg_name}, K={K}, T={T}, batch {B}/GPU, {args.mode} step ({what})',
                           step=args.mode, global_batch=B * world, slots=K, iters=T, img_size=S, hip_graph=bool(args.graph),
                           **({'options': args.option} if args.option else {}),
                           conv_path=('split_fp16x3: fp32 tensors, conv operands split on the fly into fp16 hi+lo with one power-of-two scale per 8x16 cell, '
                                      'three f16 MFMAs, fp32 accumulate - products carry >= 22 bits relative to the CELL maximum (tile-relative, not '
                                      'element-relative: profiles/r04_split_cell_stats.md); library default; `exact_fp32` on this line is the same step on fp32 MFMA'
                                      if args.conv_precision == 1 else 'exact_fp32: fp32 MFMA, v_mfma_f32_16x16x4_f32 / 32x32x2_f32 (option conv_precision=0)'),
                           parallelism=f'dp{world} (images sharded over ranks; '
                                       f'{"one RCCL all-reduce of the flat gradient buffer per step" if args.mode == "train" else "no data-path collective"})'),
               batch_iters_per_s=round(T / (dt / args.steps), 3), roofline=roofline, roofline_hbm=roofline_hbm, kernels=prof)
    if sustained:
        out['sustained'] = sustained
    if other:
        out['inference_step'] = other
        out['infer_ms_per_step'] = other['ms_per_step']
        out['infer_roofline_frac'] = other['roofline']['frac']

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
        replicas_after = parallel.replicas_identical(model.parameters())
        # north_star: "all-reduce of the per-image ELBO / gradient terms" - the (T+1, 3) {ELBO, KL, LL} batch means of the last step,
        # averaged over the ranks' equal shards (= the global-batch means; DataParallel's gather + loss.mean(), train.py:61)
        terms = parallel.allreduce_mean(model.elbo_terms.detach().float().contiguous(), world)
        out['rccl'] = dict(world_size=dist.get_world_size(), backend=dist.get_backend(), allreduce_bytes=nflat * 4,
                           allreduce_us=round(us, 1), per_step=1 if args.mode == 'train' else 0,
                           ms_per_step_by_rank=[round(v, 3) for v in rank_ms], ms_per_step_min=round(min(rank_ms), 3),
                           ms_per_step_max=round(max(rank_ms), 3),
                           replicas_identical=bool(replicas_before and replicas_after),
                           replicas_identical_before_first_step=bool(replicas_before),
                           replicas_identical_after_last_step=bool(replicas_after),
                           hsa_ipc_mode_legacy=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'),
                           elbo_terms_global_mean=[[round(float(v), 4) for v in row] for row in terms.cpu().tolist()],
                           elbo_terms_columns=['elbo', 'kl', 'log_likelihood'])
        if share:
            # test hook: all ranks drive ONE device and the collectives ran over gloo - this line is NOT a multi-GPU measurement
            out['shared_device'] = True
            out['n_devices'] = 1
            out['scaling'] = 'none (ranks share one device: test hook IODINE_BENCH_SHARE_DEVICE=1)'

    if rank == 0 and world == 1 and args.mode == 'train' and not args.no_exact_fp32 and args.conv_precision == 1:
        # The strict-precision number, first class: the same step with every conv on the exact fp32-MFMA path
        # (weight-stationary fp32 MFMA kernels since round 5; SURVEY 6c's primary plan), --steps timed steps (>= 8), its own value and roofline object with the
        # dominant launches bracketed by HIP events inside ITS timed region, priced against the 157.3 TF/s fp32 matrix peak.
        model.set_option('conv_precision', 0)
        step()
        nx = max(args.steps, 8)
        model.set_option('profile', 1)
        dtx = timed(step, nx) / nx
        model.set_option('profile', 0)
        px = read_prof()
        xm = sum(px[c]['ms_total'] for c in DOMINANT if c in px)
        xn = sum(px[c]['launches'] for c in DOMINANT if c in px)
        xall = sum(px[c].get('launches_seen', px[c]['launches']) for c in DOMINANT if c in px)
        xa = flops_per_launch / (xm / xn * 1e-3) / 1e12 if xn else 0.0
        # HBM bytes per launch of the strict kernels: their own PMC record (profiles/rNN_pmc_strict.json: the passes of tools/pmc_traffic.sh over
        # --conv-precision 0), quoted while its source digest matches the tree
        xper, xpipe, xsrc = pmc_record(args, B, K, strict=True)
        xtraffic = None
        if xper:
            xw = {c: px[c].get('launches_seen', px[c]['launches']) for c in DOMINANT if c in px}
            if xw and all(c in xper for c in xw):
                xtraffic = sum(xper[c] * n for c, n in xw.items()) / sum(xw.values())
        out['exact_fp32'] = dict(
            metric='refinement_iters_per_s', value=round(world * B * T / dtx, 2), unit='image-refinement-iters/s',
            ms_per_step=round(dtx * 1e3, 3), steps=nx, dtype='f32 (v_mfma_f32_16x16x4_f32 / 32x32x2_f32: IEEE fp32 products, fp32 accumulate)',
            vs_baseline=round(world * B * T / dtx / PUBLISHED_TRAIN_ITERS_PER_S, 3) if args.config == 'clevr6' else None,
            roofline=dict(bound='mfma',
                          kernel=FP32_KNAME.format(C=C_),
                          achieved=round(xa, 2), peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s', frac=round(xa / PEAK_F32_MFMA_TFLOPS, 4),
                          traffic=xtraffic, traffic_source=xsrc, matrix_pipe_pmc=xpipe,
                          algorithmic_hbm_bytes_per_launch=2.0 * B * K * S * S * C_ * 4,
                          flops_per_launch=flops_per_launch, avg_launch_ms=round(xm / max(xn, 1), 4), launches=xall,
                          launches_timed=xn, events_in_timed_region=True,
                          kernel_time_share=round(xm / max(xn, 1) * xall / max(nx, 1) / (dtx * 1e3), 4),
                          per_form={c: dict(ms_avg=px[c]['ms_avg'],
                                            tflops=round(flops_per_launch / (px[c]['ms_avg'] * 1e-3) / 1e12, 1),
                                            frac=round(flops_per_launch / (px[c]['ms_avg'] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4))
                                    for c in DOMINANT if c in px}),
            speedup_of_default_path=round(dtx * 1e3 / ms_per_step, 2),
            note='option conv_precision=0 (--conv-precision 0 makes it the headline line): every conv of the step on fp32 MFMA; the '
                 'default line above keeps fp32 tensors and splits conv operands into fp16 hi+lo (3 f16 MFMAs, fp32 accumulate)')
        model.set_option('conv_precision', 1)
        out['exact_fp32_ms_per_step'] = out['exact_fp32']['ms_per_step']
        out['exact_fp32_roofline_frac'] = out['exact_fp32']['roofline']['frac']

    if rank == 0 and world == 1 and not args.no_extra_configs and args.config == 'clevr6' and args.conv_precision == 1 \
            and not (args.slots or args.iters) and args.mode == 'train':
        # BASELINE configs[1] and the per-GPU shard of configs[4] on the same box, same process: 2 warm-up + 5 timed steps of each
        # step type.  Side measurements - `value` above is configs[2].
        side = {}
        for tag, (cfgname, slots, iters, bsz) in dict(cfg2=('dsprites', 6, 5, 32), cfg5_shard=('clevr6', 11, 7, 8)).items():
            m2, a2, _ = build_model(cfgname, slots, iters, device)
            cfg_of[id(a2)] = cfgname if (cfgname, slots) == ('dsprites', 6) else 'no-record'
            m2.manual_seed(99)
            x2 = torch.from_numpy(synth.make_images(bsz, a2.IMG_SIZE, seed=0)).to(device)
            rec = dict(workload=f'{"multi-dSprites 64x64" if cfgname == "dsprites" else "CLEVR 128x128"}, K={slots}, T={iters}, '
                                f'batch {bsz}, 1 GPU', steps=5)
            for md in ('train', 'infer'):
                st2, _ = make_step(m2, x2, md)
                for _ in range(2):
                    st2()
                d2 = timed(st2, 5) / 5
                rec[f'{md}_ms'] = round(d2 * 1e3, 3)
                rec[f'{md}_iters_per_s'] = round(bsz * iters / d2, 1)
                try:
                    rec['roofline' if md == 'train' else 'roofline_infer'] = side_roofline(m2, a2, bsz, st2, d2 * 1e3, md)
                except Exception as e:              # noqa: BLE001 - a side measurement must not break the bench line
                    rec['roofline' if md == 'train' else 'roofline_infer'] = dict(error=f'{type(e).__name__}: {e}')
                del st2
            side[tag] = rec
            del m2, x2
            torch.cuda.empty_cache()
        # The reference's DEFAULT decoder kernel size (lib/config/defaults.py:100, configs/test.yaml:40,44: DEC.KERNEL_SIZE 5) at the
        # CLEVR shapes: runs on the library's generic path (kernels_generic.hip: exact-fp32 MFMA convs with an LDS-resident weight slice
        # for the stride-1 decoder since round 4, scalar kernels for the stride-2 refinement stack, the spatial broadcast
        # materialised).  On the record so that the cost of that path is a number, not a guess; no roofline claim.
        try:
            from iodine_amd import IODINE
            from iodine_amd.model import arch_namespace
            a5 = arch_namespace(64, 5, 7, 128, (64, 4, 256), (64, 4), kernels=(3, 5))
            m5 = IODINE(a5)
            sh5 = {k: tuple(v.shape) for k, v in m5.state_dict().items()}
            m5.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(sh5, seed=0).items()})
            m5 = m5.to(device)
            m5.manual_seed(7)
            b5 = 4
            x5 = torch.from_numpy(synth.make_images(b5, 128, seed=0)).to(device)
            rec = dict(workload=f'CLEVR6 128x128 shapes with DEC.KERNEL_SIZE 5 (64 channels), K=7, T=5, batch {b5}, 1 GPU', steps=2,
                       path='generic decoder (kernels_generic.hip, kernels_genl0.hip): exact-fp32 MFMA convs (16x16x4 / 32x32x2), broadcast layer from per-tap '
                            'latent products; tuned refinement kernels (REF.KERNEL_SIZE 3)')
            for md in ('train', 'infer'):
                st5, _ = make_step(m5, x5, md)
                st5()
                d5 = timed(st5, 2) / 2
                rec[f'{md}_ms'] = round(d5 * 1e3, 3)
                rec[f'{md}_iters_per_s'] = round(b5 * 5 / d5, 1)
                del st5
            # the same batch on the tuned path (KERNEL_SIZE 3): what the fallback costs relative to the MFMA kernels
            m3, _, _ = build_model('clevr6', 7, 5, device)
            m3.manual_seed(7)
            for md in ('train', 'infer'):
                st3, _ = make_step(m3, x5, md)
                st3()
                d3 = timed(st3, 2) / 2
                rec[f'{md}_ms_kernel_size_3_same_batch'] = round(d3 * 1e3, 3)
                del st3
            rec['slowdown_vs_kernel_size_3'] = dict(train=round(rec['train_ms'] / rec['train_ms_kernel_size_3_same_batch'], 1),
                                                    infer=round(rec['infer_ms'] / rec['infer_ms_kernel_size_3_same_batch'], 1))
            side['default_dec_kernel5'] = rec
            del m5, m3, x5
            torch.cuda.empty_cache()
        except Exception as e:                      # noqa: BLE001 - a side measurement must not break the bench line
            side['default_dec_kernel5'] = dict(error=f'{type(e).__name__}: {e}')
        # the reference's configs/test.yaml architecture: KERNEL_SIZE 5 in BOTH stacks (decoder 32 ch x 5 layers, refinement 32 ch x 3 layers
        # stride 2), 64 x 64, K = 6, T = 5, 4-entry ENCODING - the one shipped configuration whose refinement stack is off the tuned path
        try:
            from iodine_amd.model import arch_namespace
            at = arch_namespace(16, 5, 6, 64, (32, 3, 128), (32, 5), sigma=0.14, kernels=(5, 5),
                                encoding=['posterior', 'grad_post', 'image', 'leave_one_out_likelihood'])
            mt = IODINE(at)
            sht = {k: tuple(v.shape) for k, v in mt.state_dict().items()}
            mt.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(sht, seed=0).items()})
            mt = mt.to(device)
            mt.manual_seed(7)
            bt = 32
            xt = torch.from_numpy(synth.make_images(bt, 64, seed=0)).to(device)
            rec = dict(workload=f'configs/test.yaml architecture (REF / DEC KERNEL_SIZE 5, 32 channels, 64x64, K=6, T=5), batch {bt}, 1 GPU', steps=4,
                       path='generic path, both stacks on exact-fp32 MFMA: decoder kernels_generic.hip / kernels_genl0.hip, stride-2 refinement convs '
                            'kernels_gens2.hip (v_mfma_f32_16x16x4_f32; scalar tier until round 5: 118 / 27.7 ms)')
            for md in ('train', 'infer'):
                stt, _ = make_step(mt, xt, md)
                stt()
                dt_ = timed(stt, 4) / 4
                rec[f'{md}_ms'] = round(dt_ * 1e3, 3)
                rec[f'{md}_iters_per_s'] = round(bt * 5 / dt_, 1)
                del stt
            side['test_yaml_arch'] = rec
            del mt, xt
            torch.cuda.empty_cache()
        except Exception as e:                      # noqa: BLE001
            side['test_yaml_arch'] = dict(error=f'{type(e).__name__}: {e}')
        out['configs'] = side
        for tag in ('cfg2', 'cfg5_shard'):
            for md, key in (('train', 'roofline'), ('infer', 'roofline_infer')):
                r = side.get(tag, {}).get(key) or {}
                if 'frac' in r:
                    out[f'{tag}_{md}_ms'] = side[tag][f'{md}_ms']
                    out[f'{tag}_{md}_roofline'] = dict(bound=r['bound'], frac=r['frac'], frac_mfma=r['frac_mfma'], frac_hbm=r['frac_hbm'])

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
        # the one-number summaries first (a reader of a truncated tail still finds every config's ms and frac), the long objects behind them
        head = ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data']
        summ = [k for k in out if k.endswith(('_ms_per_step', '_roofline_frac', '_train_ms', '_infer_ms', '_train_roofline', '_infer_roofline'))]
        summary = {k: out[k] for k in head if k in out}
        summary['roofline_frac'] = out['roofline']['frac']
        summary.update({k: out[k] for k in summ})
        summary.update({k: v for k, v in out.items() if k not in summary})
        print(json.dumps(summary), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
