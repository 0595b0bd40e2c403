"""world_size-2 gloo test of the data-parallel path on CPU: shard the batch over ranks, compute each shard's
loss / gradients (with the CPU oracle standing in for the device step), average with
iodine_amd.parallel.allreduce_gradients and compare with the unsharded step."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from iodine_amd import parallel
from util import golden_setup, load_golden, rel_l2


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret, flat_layout=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import iodine_oracle as O
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    lo, hi = parallel.shard_range(x.shape[0], rank, world)
    out, grads = O.train_step_grads(x[lo:hi], eps[:, lo:hi].contiguous(), params, arch)
    ps = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    if flat_layout:                      # the layout IODINE.backward produces: consecutive views of one buffer
        flat = torch.cat([grads[k].reshape(-1) for k in ps])
        off = 0
        for k, p in ps.items():
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        assert parallel._shared_flat_view([p.grad for p in ps.values()]) is not None
    else:
        for k, p in ps.items():
            p.grad = grads[k].clone()
        assert parallel._shared_flat_view([p.grad for p in ps.values()]) is None
    assert parallel.replicas_identical(ps.values())
    parallel.allreduce_gradients(ps.values(), world)
    loss = parallel.allreduce_mean(out['loss'].detach().reshape(1), world)
    # a replica that drifted is detected, and broadcast_parameters repairs it
    first = next(iter(ps.values()))
    if rank == 1:
        with torch.no_grad():
            first.view(-1)[0] += 1e-3
    assert not parallel.replicas_identical(ps.values())
    holder = torch.nn.ParameterList(list(ps.values()))
    parallel.broadcast_parameters(holder, 0)
    assert parallel.replicas_identical(ps.values())
    # same total bytes and tensor count, different per-tensor sizes: a verdict (False), not mismatched collectives
    odd = [torch.zeros(3 if rank == 0 else 5), torch.zeros(5 if rank == 0 else 3)]
    assert not parallel.replicas_identical(odd)
    # a rank that holds nothing still takes part in the count exchange
    assert not parallel.replicas_identical([] if rank == 0 else [torch.zeros(2)])
    assert parallel.replicas_identical([])
    # 2-byte dtypes: compared byte by byte (widened to int32: RCCL / gloo have no int16 MIN everywhere)
    half = [torch.full((7,), 1.5, dtype=torch.float16), torch.arange(4, dtype=torch.bfloat16)]
    assert parallel.replicas_identical(half)
    if rank == 1:
        half[0][3] = 1.501
    assert not parallel.replicas_identical(half)
    if rank == 0:
        ret['loss'] = float(loss.item())
        ret['grads'] = {k: p.grad.numpy().copy() for k, p in ps.items()}
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('flat_layout', [False, True])
def test_two_rank_gradient_average_equals_full_batch(flat_layout):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, flat_layout), nprocs=world, join=True)
    g = load_golden('tiny')
    ref_loss = float(g['f32.train.loss'])
    assert abs(ret['loss'] - ref_loss) <= 1e-5 * abs(ref_loss)
    for k, v in ret['grads'].items():
        assert rel_l2(v, g['f32.train.grad.' + k]) < 1e-4, k


def test_shard_range():
    assert parallel.shard_range(256, 3, 8) == (96, 128)
    assert [parallel.shard_range(32, r, 4) for r in range(4)] == [(0, 8), (8, 16), (16, 24), (24, 32)]
    try:
        parallel.shard_range(10, 0, 4)
        assert False
    except ValueError:
        pass
