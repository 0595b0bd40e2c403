"""One process per GPU: start N ranks of a script on this node (replaces the single-process
``torch.nn.DataParallel(model, device_ids)`` of lib/modeling/build.py:11-12).

``spawn(script, argv, nproc)`` runs ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 --master-port <free> script argv...`` and returns its exit code; the ranks read RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment (bench.py, iodine_amd.engine).  The rendezvous address is always 127.0.0.1
(a container hostname may not resolve).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def under_launcher() -> bool:
    return 'WORLD_SIZE' in os.environ and 'RANK' in os.environ


def spawn(script: str, argv, nproc: int, env=None, capture=False):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), script, *argv]
    e = dict(os.environ if env is None else env)
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: required for RCCL / cross-process device memory here
    e.setdefault('OMP_NUM_THREADS', '1')
    if capture:
        return subprocess.run(cmd, env=e, capture_output=True, text=True)
    return subprocess.run(cmd, env=e).returncode
