"""Fused Adam over all parameters in one launch.

Mirrors ``make_optimizer`` of the reference (lib/solver/build.py:5-16): ``torch.optim.Adam`` over every parameter
with ``lr = TRAIN.BASE_LR`` and ``weight_decay = TRAIN.WEIGHT_DECAY``; ``optimizer.step()`` is lib/engine/train.py:65.
The state (``exp_avg``, ``exp_avg_sq``, ``step``) and the param-group layout (one group per parameter) are torch's /
the reference's, so ``state_dict()`` round-trips with the ``'optimizer'`` entry of a reference checkpoint
(lib/utils/checkpoint.py:36-54; tests/golden/ckpt_tiny is one written by the reference).  No CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._tables = {}

    def _table(self, gi, ps):
        """(ptrs, offsets, total) device tables for one fused bucket; rebuilt when any address changes."""
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr(),
                     self.state[p]['exp_avg_sq'].data_ptr(), p.numel()) for p in ps)
        cached = self._tables.get(gi)
        if cached is None or cached[0] != key:
            dev = ps[0].device
            ptrs, offs, tot = [], [], 0
            for k in key:
                ptrs.extend(k[:4]); offs.append(tot); tot += k[4]
            cached = (key, torch.tensor(ptrs, dtype=torch.int64).to(dev), torch.tensor(offs, dtype=torch.int64).to(dev), tot)
            self._tables[gi] = cached
        return cached[1], cached[2], cached[3]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = _lib.lib()
        # The reference builds ONE PARAM GROUP PER PARAMETER (lib/solver/build.py:10-14); groups that share their
        # hyper-parameters and step count are fused into a single launch (one launch per step for the reference layout).
        buckets = {}
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.device.type != 'cuda' or p.dtype != torch.float32:
                    raise RuntimeError('FusedAdam needs float32 parameters on a ROCm device (no CPU fallback)')
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if not st:
                    st['step'] = torch.tensor(0.0)            # torch.optim.Adam keeps the step as a CPU float32 tensor
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                key = (group['lr'], tuple(group['betas']), group['eps'], group['weight_decay'], int(st['step']), p.device)
                buckets.setdefault(key, []).append(p)
        for key, ps in buckets.items():
            lr, (b1, b2), eps, wd, t0, dev = key
            t = t0 + 1
            ptrs, offs, total = self._table(key[:4] + (dev,), ps)
            with torch.cuda.device(dev):
                rc = L.iodine_adam_step(C.c_void_p(torch.cuda.current_stream().cuda_stream), _lib.ptr(ptrs), _lib.ptr(offs),
                                        len(ps), total, lr, b1, b2, eps, wd, t)
            _lib.check(rc, None, 'iodine_adam_step')
            for p in ps:
                st = self.state[p]
                st['step'] = st['step'] + 1 if torch.is_tensor(st['step']) else t     # int steps: checkpoints of old torch
                # the kernel wrote through raw pointers: tell torch (and IODINE._sync_params, which re-packs the weights
                # when a parameter's version counter moves) that the tensor changed in place
                torch.autograd.graph.increment_version(p)
        return loss


def make_optimizer(model, base_lr=3e-4, weight_decay=0.0):
    """lib/solver/build.py:5-16: Adam with one param group per parameter (``params += [{'params': [value], 'lr': lr,
    'weight_decay': weight_decay}]``), so ``optimizer.state_dict()`` has the reference's layout and the ``'optimizer'``
    entry of a reference checkpoint (lib/utils/checkpoint.py:36-54) loads with ``load_state_dict``."""
    params = [{'params': [p], 'lr': base_lr, 'weight_decay': weight_decay}
              for _, p in model.named_parameters() if p.requires_grad]
    return FusedAdam(params, lr=base_lr)
