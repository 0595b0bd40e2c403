"""Reference-shaped training / evaluation loops around the HIP module (the callers either side of the hot path,
SURVEY.md section 8f): `lib/engine/train.py:44-108`, `lib/engine/eval.py:14-28`, with one process per GPU instead of
`torch.nn.DataParallel` (`lib/modeling/build.py:11-12`).

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m iodine_amd.engine --steps 200

runs on synthetic blob scenes (there are no datasets on the box); `--clevr DIR` / `--dsprites DIR` read the reference's
dataset layouts through `iodine_amd.data`.
"""
import argparse
import os
import time

import torch

from . import parallel, synth
from .ari import ARIEvaluator
from .checkpoint import load_checkpoint, save_checkpoint
from .optim import make_optimizer


class SyntheticScenes(torch.utils.data.Dataset):
    """Deterministic blob scenes with ground-truth masks (image index = seed), same item format as lib/data/clevr.py."""

    def __init__(self, n, img_size, seed=0):
        self.n, self.s, self.seed = n, img_size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        imgs, masks = synth.make_images(1, self.s, seed=self.seed + i, kind='blobs')
        return torch.from_numpy(imgs[0]), torch.from_numpy(masks[0].astype('float32'))


def train(model, optimizer, dataloader, device, max_steps, print_every=10, checkpoint_path=None, log=print):
    """train.py:44-108: loss = model(data); loss.mean(); zero_grad; backward; [all-reduce]; step.  Returns the losses."""
    model.train()
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    losses, step, epoch = [], 0, 0
    while step < max_steps:
        sampler = getattr(dataloader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(epoch)                                         # a new shuffle per pass over the data
        epoch += 1
        for data in dataloader:
            start = time.perf_counter()
            x = data[0].to(device, non_blocking=True)                        # "first one is image" (train.py:49)
            loss = model(x).mean()
            optimizer.zero_grad()
            loss.backward()
            if world > 1:
                parallel.allreduce_gradients(model.parameters())            # replaces DataParallel's reduce
            optimizer.step()
            losses.append(loss.item())
            step += 1
            if step % print_every == 0:
                log('iter: {}, loss: {:.4f}, batch-time: {:.4f}s, lr: {}'.format(
                    step, losses[-1], time.perf_counter() - start, optimizer.param_groups[0]['lr']))
            if step >= max_steps:
                break
    if checkpoint_path and (not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0):
        save_checkpoint(checkpoint_path, model, optimizer, epoch=0, iteration=step)
    return losses


def evaluate(model, dataloader, device, evaluator=None):
    """eval.py:14-28 with the ARI evaluator of lib/eval/ari_eval.py (works under no_grad, unlike the reference).  With
    several ranks every rank evaluates its shard; the samples DistributedSampler appended to pad the shards to equal length
    are dropped (they duplicate the first images) and ``evaluator.global_mean`` holds the ARI over the whole dataset."""
    evaluator = evaluator or ARIEvaluator()
    evaluator.reset()
    model.eval()
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    world = torch.distributed.get_world_size() if dist_on else 1
    rank = torch.distributed.get_rank() if dist_on else 0
    n_total = len(dataloader.dataset)
    mine = len(range(rank, n_total, world)) if world > 1 else n_total       # un-padded share of this rank (sampler stride = world)
    with torch.no_grad():
        for image, masks in dataloader:
            evaluator.evaluate(model, (image.to(device), [m.numpy() for m in masks]))
    del evaluator.aris[mine:]
    stats = torch.tensor([float(sum(evaluator.aris)), float(len(evaluator.aris))], dtype=torch.float64,
                         device=device if dist_on and torch.distributed.get_backend() == 'nccl' else 'cpu')
    if world > 1:
        torch.distributed.all_reduce(stats)
    evaluator.global_mean = float(stats[0] / stats[1]) if float(stats[1]) > 0 else 0.0
    return evaluator


def main(argv=None):
    from . import IODINE
    from .data import CLEVR, MultiDSprites, make_dataloader
    from .model import clevr6_arch, dsprites_arch
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', choices=['clevr6', 'dsprites'], default='dsprites')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU')
    ap.add_argument('--lr', type=float, default=3e-4)                        # configs/clevr6_prop.yaml:19
    ap.add_argument('--clevr'); ap.add_argument('--dsprites')
    ap.add_argument('--resume'); ap.add_argument('--save')
    args = ap.parse_args(argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank, local = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group('nccl', rank=rank, world_size=world)    # RCCL on ROCm
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    arch = clevr6_arch() if args.config == 'clevr6' else dsprites_arch()
    torch.manual_seed(0)                                                     # same initial replica on every rank
    model = IODINE(arch).to(device)
    model.manual_seed(1000 + rank)                                           # ... but its own reparameterisation noise
    optimizer = make_optimizer(model, base_lr=args.lr)
    if args.resume:
        load_checkpoint(args.resume, model, optimizer)
    if args.clevr:
        ds = CLEVR(args.clevr)
    elif args.dsprites:
        ds = MultiDSprites(args.dsprites)
    else:
        ds = SyntheticScenes(args.batch * world * 8, arch.IMG_SIZE)
    dl = make_dataloader(ds, args.batch, shuffle=True, rank=rank, world_size=world)
    losses = train(model, optimizer, dl, device, args.steps, checkpoint_path=args.save,
                   log=print if rank == 0 else (lambda *a: None))
    ev = evaluate(model, make_dataloader(ds, args.batch, shuffle=False, rank=rank, world_size=world), device)
    if rank == 0:
        print('first loss {:.2f} -> last loss {:.2f}; Ari over all ranks: {}'.format(losses[0], losses[-1], ev.global_mean))


if __name__ == '__main__':
    main()
