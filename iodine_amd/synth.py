"""Deterministic synthetic inputs for the IODINE refinement path.

The reference ships neither datasets nor checkpoints (``.gitignore:3-5``), so
benchmarks, parity tests and golden fixtures all draw images, weights and the
reparameterisation noise from the counter-based generator below.  It is pure
integer hashing (splitmix64 finaliser) evaluated in numpy, so the build
container, the GPU box, the CPU oracle and the HIP path all see the same bytes.

Shapes / conventions follow the reference:
  * images: float32 NCHW in [0, 1] (lib/data/clevr.py:26-31, lib/data/dsprite.py:21-28)
  * weights: torch's default init bounds (kaiming-uniform a=sqrt(5) -> U(+-1/sqrt(fan_in));
    LSTMCell U(+-1/sqrt(hidden))), lib/modeling/iodine.py:54-57 leaves the defaults on
  * noise: standard normal, shape (T+1, B, K, L), one slice per ``Gaussian.sample`` call
    (lib/modeling/iodine.py:620-634)
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(v: np.ndarray) -> np.ndarray:
    v = (v + np.uint64(0x9E3779B97F4A7C15)) & _M64
    v = ((v ^ (v >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    v = ((v ^ (v >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return v ^ (v >> np.uint64(31))


def uniform01(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """n float64 uniforms in [0, 1); element i depends only on (seed, stream, i)."""
    with np.errstate(over='ignore'):
        key = _splitmix(np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix(idx ^ key)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(shape, seed: int, stream: int = 0) -> np.ndarray:
    """Standard normals (Box-Muller on two independent uniform streams), float32."""
    n = int(np.prod(shape))
    u1 = uniform01(n, seed, 2 * stream)
    u2 = uniform01(n, seed, 2 * stream + 1)
    r = np.sqrt(-2.0 * np.log1p(-u1))          # 1-u1 in (0, 1]
    return (r * np.cos(2.0 * math.pi * u2)).astype(np.float32).reshape(shape)


def make_eps(T: int, B: int, K: int, L: int, seed: int = 1) -> np.ndarray:
    """(T+1, B, K, L) noise; image b's rows depend only on (seed, b) so that a batch
    sharded over ranks sees the same noise as the unsharded batch."""
    out = np.empty((T + 1, B, K, L), dtype=np.float32)
    for b in range(B):
        out[:, b] = normal((T + 1, K, L), seed, stream=b)
    return out


def make_images(B: int, S: int, seed: int = 0, kind: str = 'uniform',
                max_objects: int = 6, first_index: int = 0):
    """Images (B, 3, S, S) float32 in [0,1].

    kind='uniform': U[0,1) per element (throughput runs; the path has no
    data-dependent control flow).  kind='blobs': grey background with 3..max_objects
    filled discs / squares in painter's order; also returns ground-truth masks
    (list of (n_obj, S, S) uint8, background excluded like lib/data/clevr.py:72).
    Image b depends only on (seed, first_index + b)."""
    imgs = np.empty((B, 3, S, S), dtype=np.float32)
    masks = []
    yy, xx = np.meshgrid(np.arange(S), np.arange(S), indexing='ij')
    for b in range(B):
        gi = first_index + b
        if kind == 'uniform':
            imgs[b] = uniform01(3 * S * S, seed, stream=gi).astype(np.float32).reshape(3, S, S)
            continue
        u = uniform01(8 * (max_objects + 1), seed + 7919, stream=gi)
        n_obj = 3 + int(u[0] * (max_objects - 2)) if max_objects > 3 else max_objects
        n_obj = min(n_obj, max_objects)
        img = np.full((3, S, S), 0.25, dtype=np.float32)
        owner = np.zeros((S, S), dtype=np.int32)
        for o in range(n_obj):
            v = u[8 * (o + 1): 8 * (o + 2)]
            cx, cy = v[0] * S, v[1] * S
            rad = S / 16.0 + v[2] * (S / 6.0 - S / 16.0)
            col = v[3:6].astype(np.float32)
            if v[6] < 0.5:
                sel = (xx - cx) ** 2 + (yy - cy) ** 2 <= rad * rad
            else:
                sel = (np.abs(xx - cx) <= rad) & (np.abs(yy - cy) <= rad)
            img[:, sel] = col[:, None]
            owner[sel] = o + 1
        imgs[b] = img
        masks.append(np.stack([(owner == o + 1) for o in range(n_obj)]).astype(np.uint8))
    return (imgs, masks) if kind == 'blobs' else imgs


def make_params(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, dec_gain: float = 1.0,
                posterior_scale: float = 0.0, lstm_hidden: int = None) -> Dict[str, np.ndarray]:
    """Weights with torch's default-init bounds, generated name by name.

    dec_gain scales decoder conv weights (an untrained net with gain 1 produces nearly
    flat outputs; tests use ~3 to exercise the non-linearities).  posterior_scale != 0
    gives ``posterior.init_*`` non-zero values so their gradient path is exercised."""
    out = {}
    for j, (name, shp) in enumerate(shapes.items()):
        n = int(np.prod(shp))
        u = uniform01(n, seed + 104729, stream=j)
        if name.startswith('posterior.'):
            w = (2 * u - 1) * posterior_scale
        elif name.startswith('refine.lstm.'):
            H = lstm_hidden if lstm_hidden is not None else shapes['refine.lstm.weight_hh'][1]
            w = (2 * u - 1) / math.sqrt(H)
        else:
            wname = name.rsplit('.', 1)[0] + '.weight'
            wshape = shapes[wname]
            fan_in = int(np.prod(wshape[1:]))
            w = (2 * u - 1) / math.sqrt(fan_in)
            if name.startswith('decoder.') and name.endswith('.weight'):
                w = w * dec_gain
        out[name] = w.astype(np.float32).reshape(shp)
    return out
