"""CPU restatement of the reference's ARI metric (test infrastructure).

Follows lib/utils/ari.py:6-54 (contingency table -> adjusted Rand index through
pair counts) and lib/eval/ari_eval.py:22-39 (argmax over K -> one-hot masks).
Integer arithmetic: n-choose-2 is evaluated exactly as n*(n-1)/2 in float64
(what ``scipy.special.comb(n, 2)`` returns for these magnitudes).
"""
from __future__ import annotations

import numpy as np


def _comb2(v):
    v = np.asarray(v, dtype=np.float64)
    return v * (v - 1.0) / 2.0


def compute_ari(table) -> float:
    """lib/utils/ari.py:6-33.  ``table`` is the (r, s) contingency table."""
    table = np.asarray(table)
    a = table.sum(axis=1)
    b = table.sum(axis=0)
    n = a.sum()
    comb_a = _comb2(a).sum()
    comb_b = _comb2(b).sum()
    comb_n = _comb2(n)
    comb_t = _comb2(table).sum()
    if comb_b == comb_a == comb_n == comb_t:
        return 1.0
    return float((comb_t - comb_a * comb_b / comb_n)
                 / (0.5 * (comb_a + comb_b) - (comb_a * comb_b) / comb_n))


def contingency(gt_masks, pred_masks) -> np.ndarray:
    """lib/utils/ari.py:36-52: table[i, j] = |gt_i AND pred_j| over pixels.
    gt_masks (N0, H, W), pred_masks (N1, H, W), both 0/1."""
    g = np.asarray(gt_masks).astype(bool).reshape(len(gt_masks), -1)
    q = np.asarray(pred_masks).astype(bool).reshape(len(pred_masks), -1)
    return (g[:, None, :] & q[None, :, :]).sum(axis=-1).astype(np.int64)


def compute_mask_ari(gt_masks, pred_masks) -> float:
    return compute_ari(contingency(gt_masks, pred_masks))


def binarize_argmax(mask) -> np.ndarray:
    """lib/eval/ari_eval.py:25-35: soft masks (B, K, 1, H, W) -> one-hot (B, K, H, W)."""
    m = np.asarray(mask)[:, :, 0]
    idx = m.argmax(axis=1)
    K = m.shape[1]
    return (np.arange(K)[None, :, None, None] == idx[:, None]).astype(np.float32)
