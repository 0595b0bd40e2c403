"""ARI evaluation with the argmax + contingency table computed on the device.

Mirrors ``ARIEvaluator`` (lib/eval/ari_eval.py:7-46) and ``compute_ari`` (lib/utils/ari.py:6-33): the model's soft
masks are binarised by argmax over K, the (N_gt x K) table of pixel co-occurrences is built with int32 atomics by
``iodine_ari_table``, and the pair-count formula on that tiny integer table is evaluated on the host in float64
(``n choose 2 = n (n - 1) / 2`` is exact at these magnitudes, like ``scipy.special.comb``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def compute_ari(table) -> float:
    """lib/utils/ari.py:6-33."""
    t = np.asarray(table, dtype=np.float64)
    c2 = lambda v: v * (v - 1.0) / 2.0
    a, b = t.sum(axis=1), t.sum(axis=0)
    n = a.sum()
    ca, cb, cn, ct = c2(a).sum(), c2(b).sum(), c2(n), c2(t).sum()
    if cb == ca == cn == ct:
        return 1.0
    return float((ct - ca * cb / cn) / (0.5 * (ca + cb) - (ca * cb) / cn))


def ari_tables(mask: torch.Tensor, gt_masks) -> np.ndarray:
    """mask (B,K,1,S,S) on the device, gt_masks: list of (N_b, S, S) 0/1 arrays -> int tables (B, G, K), G = max N_b."""
    if mask.device.type != 'cuda':
        raise RuntimeError('ari_tables needs the masks on a ROCm device (no CPU fallback)')
    B, K, _, S, _ = mask.shape
    G = max(int(m.shape[0]) for m in gt_masks)
    gt = torch.zeros((B, G, S, S), dtype=torch.uint8)
    for b, m in enumerate(gt_masks):
        gt[b, :m.shape[0]] = torch.as_tensor(np.asarray(m)).to(torch.uint8)
    gt = gt.to(mask.device)
    table = torch.empty((B, G, K), dtype=torch.int32, device=mask.device)
    mk = mask.detach().to(torch.float32).contiguous()
    with torch.cuda.device(mask.device):
        rc = _lib.lib().iodine_ari_table(C.c_void_p(torch.cuda.current_stream().cuda_stream), _lib.ptr(mk), _lib.ptr(gt),
                                         B, K, G, S * S, _lib.ptr(table))
    _lib.check(rc, None, 'iodine_ari_table')
    return table.cpu().numpy()


class ARIEvaluator:
    """Same protocol as lib/eval/base.py:1-11 / lib/eval/ari_eval.py:7-46: evaluate(model, data), reset(), get_results()."""

    def __init__(self):
        self.aris = []

    def evaluate(self, model, data):
        image, gt_masks = data
        pred, pred_mask, mean = model.reconstruct(image)
        tables = ari_tables(pred_mask, gt_masks)
        for b, m in enumerate(gt_masks):
            self.aris.append(compute_ari(tables[b, :m.shape[0]]))

    def reset(self):
        self.aris = []

    def get_results(self):
        return 'Ari: {}'.format(np.mean(self.aris) if self.aris else 0)
