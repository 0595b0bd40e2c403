"""Fused Adam over all parameters in one launch.

Mirrors ``make_optimizer`` of the reference (lib/solver/build.py:5-16): ``torch.optim.Adam`` over every parameter
with ``lr = TRAIN.BASE_LR`` and ``weight_decay = TRAIN.WEIGHT_DECAY``; ``optimizer.step()`` is lib/engine/train.py:65.
The state (``exp_avg``, ``exp_avg_sq``, ``step``) uses torch's names so that ``state_dict()`` round-trips with the
``'optimizer'`` entry of a reference checkpoint (lib/utils/checkpoint.py:36-54).  No CPU / eager fallback.
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
        """(ptrs, offsets, total) device tables for one param group; rebuilt when any address changes."""
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
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if p.device.type != 'cuda' or p.dtype != torch.float32:
                    raise RuntimeError('FusedAdam needs float32 parameters on a ROCm device (no CPU fallback)')
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            steps = {int(self.state[p]['step']) for p in ps}
            if len(steps) != 1:
                raise RuntimeError('FusedAdam: parameters of one group must share the step count')
            t = steps.pop() + 1
            ptrs, offs, total = self._table(gi, ps)
            b1, b2 = group['betas']
            with torch.cuda.device(ps[0].device):
                rc = L.iodine_adam_step(C.c_void_p(torch.cuda.current_stream().cuda_stream), _lib.ptr(ptrs), _lib.ptr(offs),
                                        len(ps), total, group['lr'], b1, b2, group['eps'], group['weight_decay'], t)
            _lib.check(rc, None, 'iodine_adam_step')
            for p in ps:
                self.state[p]['step'] = t
                # the kernel wrote through raw pointers: tell torch (and IODINE._sync_params, which re-packs the weights
                # when a parameter's version counter moves) that the tensor changed in place
                torch.autograd.graph.increment_version(p)
        return loss


def make_optimizer(model, base_lr=3e-4, weight_decay=0.0):
    """lib/solver/build.py:5-16 (one param group per parameter there; one group here, same arithmetic)."""
    return FusedAdam([p for p in model.parameters() if p.requires_grad], lr=base_lr, weight_decay=weight_decay)
