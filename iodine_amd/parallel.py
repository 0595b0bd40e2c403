"""Rank-per-GPU data parallelism for the IODINE step (replaces torch.nn.DataParallel,
lib/modeling/build.py:11-12).

The path shards over images only: every image is independent (layer-norms are per (b, k), the ELBO is a
batch mean of per-image sums, the inner gradients are per-image by construction - SURVEY.md section 8e), and
the K slots of one image must stay on one GPU (softmax / log-mixture / leave-one-out couple them per pixel).
DataParallel's per-step traffic (parameter broadcast, input scatter, loss gather, gradient reduce-to-device-0
on each of the T+1 backwards) collapses to ONE all-reduce per training step of the flat gradient buffer
(4.44 MB for the CLEVR architecture) over RCCL/xGMI; inference needs no collective at all.
"""
from __future__ import annotations

from typing import Iterable, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Images [start, stop) of the global batch owned by `rank` (equal shards, like DataParallel's scatter)."""
    if global_batch % world != 0:
        raise ValueError(f'global batch {global_batch} is not divisible by world size {world}')
    per = global_batch // world
    return rank * per, (rank + 1) * per


def _shared_flat_view(grads):
    """If the gradients are consecutive slices of ONE storage (the layout IODINE's backward hands to autograd: views of
    the flat buffer iodine_train_backward fills), return that storage range as a 1-D tensor; else None."""
    g0 = grads[0]
    if not all(g.is_contiguous() and g.dtype == g0.dtype and g.device == g0.device for g in grads):
        return None
    base = g0.untyped_storage().data_ptr()
    order = sorted(grads, key=lambda g: g.storage_offset())          # any parameter order, as long as the slices tile a range
    start = off = order[0].storage_offset()
    for g in order:
        if g.untyped_storage().data_ptr() != base or g.storage_offset() != off:
            return None
        off += g.numel()
    return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(g0.untyped_storage(), start, (off - start,))


def allreduce_gradients(params: Iterable[torch.nn.Parameter], world: int = None, group=None) -> None:
    """Average .grad over ranks with ONE all-reduce (== ``loss.mean()`` over DataParallel replicas,
    lib/engine/train.py:61).  Works with any backend (nccl = RCCL on ROCm, gloo on CPU).  When the gradients already
    live back to back in one buffer (they do after IODINE.backward) the collective runs on that buffer in place;
    otherwise they are gathered into a flat buffer and scattered back."""
    if not dist.is_available() or not dist.is_initialized():
        return
    world = world or dist.get_world_size(group)
    if world == 1:
        return
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    with torch.no_grad():
        flat = _shared_flat_view([p.grad for p in ps])
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            flat.div_(world)
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for p in ps:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


def allreduce_mean(t: torch.Tensor, world: int = None, group=None) -> torch.Tensor:
    """Mean of a (small) tensor over ranks: ELBO / KL / LL scalars, ARI sums."""
    if not dist.is_available() or not dist.is_initialized():
        return t
    world = world or dist.get_world_size(group)
    if world == 1:
        return t
    out = t.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out / world


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None) -> None:
    """Every rank takes rank ``src``'s parameters (DataParallel replicates device 0's module on every forward,
    lib/modeling/build.py:11-12; rank-per-GPU replicas are made identical once, before step 1, and stay identical because every
    rank applies the same all-reduced gradients).  Writes bypass autograd's version counter, so the module re-packs its weights."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for p in model.parameters():
            dist.broadcast(p.data, src, group=group)
    if hasattr(model, 'mark_params_dirty'):
        model.mark_params_dirty()


def replicas_identical(params: Iterable[torch.nn.Parameter], group=None) -> bool:
    """True iff every rank holds BITWISE the same parameters, element by element: the raw int32 bit patterns of all tensors, back
    to back, are all-reduced with MIN and with MAX (two small collectives over 4.4 MB at the CLEVR architecture); the replicas are
    identical iff min == max everywhere and the element counts agree.  (A sum of bit patterns - what this used to compare - is a
    checksum: permuted values or offsetting +1 / -1 ulp drifts pass it.)  Run before the first step and after the last."""
    if not dist.is_available() or not dist.is_initialized():
        return True
    world = dist.get_world_size(group)
    if world == 1:
        return True
    with torch.no_grad():
        ps = list(params)
        # every rank enters the SAME sequence of collectives, whatever it holds: first the tensor count (a rank without parameters
        # still takes part), then the per-tensor byte sizes, and only when both agree everywhere the bit patterns themselves
        if ps:
            dev = ps[0].device
        else:                                              # (a rank without tensors: the backend decides where collectives live)
            dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
        cnt = torch.tensor([len(ps)], dtype=torch.int64, device=dev)
        c_lo, c_hi = cnt.clone(), cnt.clone()
        dist.all_reduce(c_lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(c_hi, op=dist.ReduceOp.MAX, group=group)
        if int(c_lo.item()) != int(c_hi.item()):
            return False                                   # different numbers of tensors
        if not ps:
            return True                                    # nothing to compare, on any rank
        sizes = torch.tensor([p.numel() * p.element_size() for p in ps] + [p.element_size() for p in ps], dtype=torch.int64, device=dev)
        s_lo, s_hi = sizes.clone(), sizes.clone()
        dist.all_reduce(s_lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(s_hi, op=dist.ReduceOp.MAX, group=group)
        if not torch.equal(s_lo, s_hi):
            return False                                   # same count, different per-tensor sizes: the collectives below would not line up
        # per tensor, in chunks of at most 16 Mi elements: one reused pair of buffers instead of three copies of all parameters;
        # every rank runs the same sequence of collectives (the per-tensor sizes were just found equal); the verdict is formed at the end.
        # 4-byte dtypes are compared as int32 bit patterns; anything else byte by byte, widened to int32 (RCCL has no int16 / uint8 MIN)
        same = True
        CH = 1 << 24
        for p in ps:
            flat = p.detach().contiguous().reshape(-1)
            bits = flat.view(torch.int32) if flat.element_size() == 4 else flat.view(torch.uint8).to(torch.int32)
            for o in range(0, bits.numel(), CH):
                lo, hi = bits[o:o + CH].clone(), bits[o:o + CH].clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
                same = same and torch.equal(lo, hi)
    return bool(same)
