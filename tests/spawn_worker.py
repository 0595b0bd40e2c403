"""Worker for tests/test_bench_spawn_cpu.py: what a rank of bench.py does around its timed region, on gloo."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iodine_amd import parallel  # noqa: E402

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
p = torch.nn.Parameter(torch.zeros(5))
p.grad = torch.full((5,), float(rank + 1))
parallel.allreduce_gradients([p], world)
t = torch.tensor([float(rank)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
if rank == 0:
    print(json.dumps(dict(world=dist.get_world_size(), grad=p.grad.tolist(), max_rank=t.item(), argv=sys.argv[1:])))
dist.destroy_process_group()
