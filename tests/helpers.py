"""Shared test helpers (test infrastructure; may import the oracle)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'vae-npvc_amd')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# A shrunk architecture whose every activation fits in a few KB; exercises generic
# geometry: even kernel (asymmetric SAME pad), different strides, y_dim != 10.
SMALL_ARCH = {
    "mode": "VAE", "hwc": [54, 1, 1], "z_dim": 8, "y_dim": 3, "y_emb_dim": 8,
    "encoder": {"kernel": [[5, 1], [4, 1]], "stride": [[3, 1], [3, 1]], "output": [4, 8]},
    "generator": {"hwc": [6, 1, 5], "kernel": [[5, 1], [4, 1], [11, 1]], "stride": [[3, 1], [3, 1], [1, 1]],
                  "output": [4, 2, 1]},
    "training": {"datadir": "", "batch_size": 4, "lr": 1e-4, "beta1": 0.5, "beta2": 0.999, "max_iter": 3},
}


def load_arch():
    with open(os.path.join(PKG, 'architecture-vae-vcc2016.json')) as fp:
        return json.load(fp)


def rel_err(a, b):
    """||a-b||_inf / max(||b||_inf, 1e-6)  (SURVEY 8d parity bar)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))


def sample_idx(n, k=8, seed=12345):
    rng = np.random.Generator(np.random.PCG64(seed + n))
    return np.sort(rng.choice(n, size=min(k, n), replace=False))


def golden_large_inputs(arch, gold, F, seed):
    """The kink-safe batch of a large golden fixture (tests/golden/make_golden_large.py): frames `frame_src` of the
    seeded candidate stream."""
    from oracle import convvae_oracle as O
    xc, yc, ec = O.make_inputs(arch, int(gold['n_candidates']), seed)
    idx = np.asarray(gold['frame_src'], np.int64)
    assert idx.shape == (F,)
    return xc[idx], yc[idx], ec[idx]
