"""The N > 1 path on real devices: runs wherever the box has at least two ROCm devices (the 1-GPU boxes of the build loop skip
it; tests/test_parallel_cpu.py and tests/test_bench_spawn_cpu.py cover the same code on gloo).  One process per GPU over RCCL,
rendezvous on 127.0.0.1, HSA_ENABLE_IPC_MODE_LEGACY=0 (iodine_amd.launch.spawn)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from iodine_amd import launch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
two_devices = pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason='needs >= 2 ROCm devices')


def _last_json(text):
    return json.loads([ln for ln in text.splitlines() if ln.startswith('{')][-1])


@two_devices
def test_bench_two_ranks_over_rccl():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--no-exact-fp32'], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out['n_gpus'] == 2 and out['rccl']['world_size'] == 2 and out['rccl']['backend'] == 'nccl'
    assert out['config']['global_batch'] == 64 and out['scaling'] == 'weak'
    assert out['rccl']['replicas_identical'] is True
    assert out['value'] > 0 and out['ms_per_step'] > 0


@two_devices
def test_two_rank_gradients_equal_the_unsharded_step():
    r = launch.spawn(os.path.join(ROOT, 'tests', 'rccl_worker.py'), [], 2, capture=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out['world'] == 2 and out['backend'] == 'nccl'
    assert out['same_params'] and out['in_place']
    assert out['bitwise_equal_over_ranks']                      # every rank holds the same averaged gradients
    assert out['grad_rel_l2_vs_unsharded'] < 2e-5
    assert out['loss_rel_err_vs_unsharded'] < 1e-6 and out['loss_rel_err_vs_reference'] < 1e-4


# ---- the same two tests with BOTH ranks on one device and the collectives over gloo: runs on the 1-GPU boxes of the build loop.
# Two processes drive the HIP library on the same GPU at once (two handles, two arenas, two streams), shard the images, all-reduce the
# flat gradient buffer in place and step Adam - everything of the N > 1 path except RCCL itself (which refuses two ranks on one device).
def _shared_env():
    e = dict(os.environ)
    e['IODINE_BENCH_SHARE_DEVICE'] = '1'
    return e


def test_two_ranks_sharing_one_device_bench_path():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '8',
                        '--no-cpu-baseline', '--no-exact-fp32', '--no-sustain'], capture_output=True, text=True, timeout=1500,
                       env=_shared_env())
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out['n_gpus'] == 2 and out['rccl']['world_size'] == 2 and out['rccl']['backend'] == 'gloo'
    assert out['config']['global_batch'] == 16
    # the hook is visible on the line: two ranks on ONE device are not a multi-GPU measurement
    assert out['shared_device'] is True and out['n_devices'] == 1 and out['scaling'].startswith('none')
    assert out['rccl']['replicas_identical'] is True          # after Adam steps on all-reduced gradients the replicas still agree bitwise
    # the (T+1, 3) ELBO / KL / LL batch means of the last step, all-reduced over the ranks (north_star: "all-reduce of the per-image
    # ELBO / gradient terms"): ELBO = LL - KL row by row
    terms = out['rccl']['elbo_terms_global_mean']
    assert len(terms) in (5, 6) and all(abs(t[0] - (t[2] - t[1])) <= 1e-4 * abs(t[0]) for t in terms)   # (T elbo calls if the last step was a reconstruct, T + 1 after a training step)
    assert out['value'] > 0 and out['ms_per_step'] > 0 and out['roofline']['frac'] > 0


def test_two_ranks_sharing_one_device_gradients_equal_the_unsharded_step():
    r = launch.spawn(os.path.join(ROOT, 'tests', 'rccl_worker.py'), [], 2, env=_shared_env(), capture=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out['world'] == 2 and out['backend'] == 'gloo'
    assert out['same_params'] and out['in_place']
    assert out['bitwise_equal_over_ranks']
    assert out['grad_rel_l2_vs_unsharded'] < 2e-5
    assert out['loss_rel_err_vs_unsharded'] < 1e-6 and out['loss_rel_err_vs_reference'] < 1e-4
