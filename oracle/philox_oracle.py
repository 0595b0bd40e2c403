"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

numpy restatement of the library's counter-based normal generator (``iodine_randn``, csrc/kernels_misc.hip): the
reference draws its reparameterisation noise with ``torch.randn_like`` (lib/modeling/iodine.py:632) - the values of a
torch generator are not part of the reference's contract, so the library uses its own stream:

* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 1.09):
  counter = {quad lo, quad hi, stream lo, stream hi}, key = {seed lo, seed hi};
* the four 32-bit outputs of quad q become normals 4q .. 4q+3 through two Box-Muller pairs, uniforms
  u = ((r >> 9) + 0.5) * 2^-23 (exact in fp32, never 0 or 1).

Pinned by the Random123 known-answer vectors for philox4x32-10 (tests/test_philox_cpu.py).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(ctr, key):
    """ctr (..., 4) uint32, key (..., 2) uint32 -> (..., 4) uint32."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32).copy()
    k1 = np.asarray(key[..., 1], dtype=np.uint32).copy()
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def randn(n, seed, stream_id):
    """The n float32 normals ``iodine_randn(out, n, seed, stream_id)`` writes."""
    quads = (n + 3) // 4
    q = np.arange(quads, dtype=np.uint64)
    ctr = np.empty((quads, 4), dtype=np.uint32)
    ctr[:, 0] = (q & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (q >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = np.uint32(stream_id & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32((stream_id >> 32) & 0xFFFFFFFF)
    key = np.empty((quads, 2), dtype=np.uint32)
    key[:, 0] = np.uint32(seed & 0xFFFFFFFF)
    key[:, 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    r = philox4x32_10(ctr, key)
    u = ((r >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    out = np.empty((quads, 4), dtype=np.float32)
    for j in range(2):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[:, 2 * j]))
        ang = np.float32(6.283185307179586) * u[:, 2 * j + 1]
        out[:, 2 * j] = rad * np.cos(ang)
        out[:, 2 * j + 1] = rad * np.sin(ang)
    return out.reshape(-1)[:n]
