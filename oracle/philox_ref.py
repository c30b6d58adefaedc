"""NumPy restatement of the device sampler draw (TEST INFRASTRUCTURE).

The reference draws the sampler noise with an unseeded ``tf.random_normal``
(util/layers.py:154), so there is nothing of the reference to pin here; this file pins OUR
generator (vae-npvc_amd/csrc/philox.h): Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel
random numbers: as easy as 1, 2, 3", SC'11) + Box-Muller on 24-bit uniforms.  The block function
is checked against the published Random123 known-answer vectors in tests/test_oracle.py.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: uint32 [..., 2] -> uint32 [..., 4]."""
    c = [np.asarray(ctr[..., i], np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[..., 0], np.uint32).copy()
    k1 = np.asarray(key[..., 1], np.uint32).copy()
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def normal(n, seed, offset=0):
    """The first n elements of the device draw for (seed, offset), float32."""
    nq = (n + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    ctr = np.stack([(q & MASK).astype(np.uint32), (q >> np.uint64(32)).astype(np.uint32),
                    np.full(nq, offset & 0xFFFFFFFF, np.uint32), np.full(nq, (offset >> 32) & 0xFFFFFFFF, np.uint32)], -1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32), (nq, 2))
    x = philox4x32_10(ctr, key)
    u = ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    out = np.empty((nq, 4), np.float32)
    for a, b, o in ((0, 1, 0), (2, 3, 2)):
        r = np.sqrt(np.float32(-2.0) * np.log(u[:, a]))
        ang = np.float64(2.0) * np.pi * u[:, b].astype(np.float64)
        out[:, o] = r * np.cos(ang).astype(np.float32)
        out[:, o + 1] = r * np.sin(ang).astype(np.float32)
    return out.reshape(-1)[:n]


def uniform(n, seed, offset=0):
    """The first n elements of the device U[0,1) draw for (seed, offset), float32: word e & 3 of counter
    (e >> 2, offset), top 24 bits."""
    nq = (n + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    ctr = np.stack([(q & MASK).astype(np.uint32), (q >> np.uint64(32)).astype(np.uint32),
                    np.full(nq, offset & 0xFFFFFFFF, np.uint32), np.full(nq, (offset >> 32) & 0xFFFFFFFF, np.uint32)], -1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32), (nq, 2))
    x = philox4x32_10(ctr, key)
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(-1)[:n]
