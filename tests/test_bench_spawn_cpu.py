"""bench.py --gpus N starts its own N ranks (iodine_amd.launch.spawn -> torch.distributed.run): the spawn path on gloo,
and the clear failure of a multi-GPU request on a box without that many devices."""
import json
import os
import subprocess
import sys

import torch

from iodine_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spawn_runs_n_ranks_with_gloo():
    r = launch.spawn(os.path.join(ROOT, 'tests', 'spawn_worker.py'), ['--tag', 'x'], 2, capture=True)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['world'] == 2 and out['max_rank'] == 1.0 and out['argv'] == ['--tag', 'x']
    assert out['grad'] == [1.5] * 5                      # mean of the ranks' gradients (1 and 2)


def test_bench_multi_gpu_request_without_devices_fails_clearly():
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('box has >= 2 devices')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert 'needs 2 visible ROCm devices' in r.stderr, r.stderr[-2000:]
    assert 'AssertionError' not in r.stderr
