"""Generates the committed golden fixtures from the CPU oracle (float64).

    python tests/golden/make_golden.py

PARITY UNPINNED: the reference cannot be executed here (no TensorFlow), so these are
outputs of OUR restatement (oracle/convvae_oracle.py), generated with deterministic
seeded inputs/weights.  Fixtures store outputs only; inputs are regenerated from seeds.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from helpers import SMALL_ARCH, load_arch, sample_idx  # noqa: E402
from oracle import convvae_oracle as O  # noqa: E402


def run(arch, F, seed, full):
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    R = O.np_forward(arch, P, x, y, eps)
    L, G = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64)
    out = {'z_mu': R['z_mu'], 'z_lv': R['z_lv'], 'xh': R['xh'],
           'loss3': np.array([R['G'], R['D_KL'], R['logP']])}
    assert abs(L['G'] - R['G']) < 1e-9 * max(1, abs(R['G']))
    names = list(G.keys())
    out['grad_l2'] = np.array([np.sqrt((G[n].astype(np.float64) ** 2).sum()) for n in names])
    out['grad_absmax'] = np.array([np.abs(G[n]).max() for n in names])
    out['grad_samples'] = np.stack([np.pad(G[n].ravel()[sample_idx(G[n].size)], (0, 8 - min(8, G[n].size)))
                                    for n in names])
    # three TF-Adam steps on the same batch (float64)
    p = O.flatten_params(P).astype(np.float64)
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    idx = sample_idx(p.size, 64)
    for t in (1, 2, 3):
        Pt = O.unflatten_params(arch, p)
        _, Gt = O.torch_loss_and_grads(arch, Pt, x, y, eps, torch.float64)
        g = np.concatenate([Gt[n].ravel() for n in names]).astype(np.float64)
        p, m, v = O.tf_adam_step(p, g, m, v, t)
        out['adam_p%d' % t] = p[idx]
        if full:
            out['adam_full_p%d' % t] = p
    if full:
        for k in R:
            if k.startswith(('enc_a', 'dec_a', 'h', 'z')):
                out['act_' + k] = R[k]
        for n in names:
            out['grad_' + n.replace('/', '__')] = G[n]
    return out


if __name__ == '__main__':
    np.savez_compressed(os.path.join(HERE, 'vcc2016_F4_seed0.npz'), **run(load_arch(), 4, 0, False))
    np.savez_compressed(os.path.join(HERE, 'small_F5_seed1.npz'), **run(SMALL_ARCH, 5, 1, True))
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))
