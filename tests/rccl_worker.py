"""Rank of tests/test_gpu_multirank.py: one training step of the HIP module on this rank's shard of the cfg1 golden batch, the
in-place RCCL all-reduce of the flat gradient buffer (iodine_amd.parallel.allreduce_gradients - what replaces DataParallel's
gather, lib/modeling/build.py:11-12 / lib/engine/train.py:61), then every rank's averaged gradients are gathered on rank 0 and
compared bitwise with each other and, to summation order, with the unsharded step run on rank 0's device."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from iodine_amd import parallel  # noqa: E402
from util import golden_setup, load_golden, make_hip_model, rel_l2  # noqa: E402

rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
share = os.environ.get('IODINE_BENCH_SHARE_DEVICE') == '1'        # 1-GPU box: both ranks on device 0, collectives over gloo
dev = torch.device('cuda', 0 if share else local)
torch.cuda.set_device(dev)
if share:
    dist.init_process_group('gloo', rank=rank, world_size=world)
else:
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

g = load_golden('cfg1_dsprites_k4_t3_b4')
arch, params, x, eps, _ = golden_setup(g)
m = make_hip_model(arch, params, dev)
# replicas must start from identical parameters (DataParallel broadcasts them every step): compare a checksum over ranks
h = torch.stack([p.detach().double().sum() for p in m.parameters()]).sum().reshape(1)
hs = [torch.zeros_like(h) for _ in range(world)]
dist.all_gather(hs, h)
same_params = all(torch.equal(hs[0], t) for t in hs)

lo, hi = parallel.shard_range(x.shape[0], rank, world)
m.zero_grad(set_to_none=True)
loss = m(x[lo:hi].to(dev), eps[:, lo:hi].contiguous().to(dev))
loss.backward()
in_place = parallel._shared_flat_view([p.grad for p in m.parameters()]) is not None
parallel.allreduce_gradients(m.parameters(), world)
loss_mean = parallel.allreduce_mean(loss.detach().reshape(1), world)
flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
gathered = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
if rank == 0:
    bitwise = all(torch.equal(gathered[0], t) for t in gathered)
    m.zero_grad(set_to_none=True)
    full = m(x.to(dev), eps.to(dev))
    full.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    gold = float(g['f32.train.loss'])
    print(json.dumps(dict(world=dist.get_world_size(), backend=dist.get_backend(), same_params=same_params, in_place=in_place,
                          bitwise_equal_over_ranks=bitwise, grad_rel_l2_vs_unsharded=rel_l2(flat.cpu().numpy(), ref.cpu().numpy()),
                          loss_rel_err_vs_unsharded=abs(loss_mean.item() - full.item()) / abs(full.item()),
                          loss_rel_err_vs_reference=abs(loss_mean.item() - gold) / abs(gold))), flush=True)
dist.barrier()
dist.destroy_process_group()
