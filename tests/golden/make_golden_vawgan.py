"""Generates the committed golden fixture of the VAWGAN branch from the float64 autograd oracle.

    python tests/golden/make_golden_vawgan.py

PARITY UNPINNED (see oracle/vawgan_oracle.py: the reference tree holds the trainer of this branch, not its model).
The fixture stores outputs only; inputs and weights are regenerated from the seeds below.
  critic step     F = 4 : W_dist, gp, critic values, gradient of l_D = -W_dist + 10 gp (every tensor; the 115-tap kernel
                          as norm, largest entry and samples)
  generator step  F = 4 : losses, gradients of l_E ('Encoder') and l_G = -logP + 50 W_dist ('Generator', 'y_emb')
                          as norm, largest entry and samples per tensor
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from helpers import PKG, sample_idx  # noqa: E402
from oracle import convvae_oracle as O  # noqa: E402
from oracle import vawgan_oracle as V  # noqa: E402

F, SEED_P, SEED_D, SEED_X, LAM, ALPHA = 4, 21, 22, 23, 10.0, 50.0


def inputs(arch):
    x, y, eps = O.make_inputs(arch, F, SEED_X)
    rng = np.random.RandomState(SEED_X + 1)
    xh = np.tanh(0.7 * rng.randn(F, arch['hwc'][0]) + 0.2)
    u = rng.rand(F)
    return x, y, eps, xh, u


def summarise(G):
    names = list(G.keys())
    return {'grad_l2': np.array([np.sqrt((np.asarray(G[n], np.float64) ** 2).sum()) for n in names]),
            'grad_absmax': np.array([np.abs(G[n]).max() for n in names]),
            'grad_samples': np.stack([np.pad(np.asarray(G[n]).ravel()[sample_idx(G[n].size)], (0, 8 - min(8, G[n].size)))
                                      for n in names])}


def run():
    with open(os.path.join(PKG, 'architecture-vawgan-vcc2016.json')) as fp:
        arch = json.load(fp)
    P, D = O.init_params(arch, SEED_P), V.disc_init_params(arch, SEED_D)
    x, y, eps, xh, u = inputs(arch)
    out = {}
    L, G = V.critic_loss_and_grads(arch, D, x, xh, u, LAM)
    out.update({'critic_' + k: v for k, v in summarise(G).items()})
    out['critic_losses'] = np.array([L['W_dist'], L['gp'], L['l_D']])
    out['critic_values'] = np.concatenate([L['d_real'], L['d_fake'], L['d_int']])
    out['critic_input_grad_norm'] = L['norm']
    for k in G:
        if G[k].size <= 4096:
            out['critic_grad_' + k.replace('/', '__')] = G[k]
    L2, G2 = V.encoder_generator_grads(arch, P, D, x, y, eps, ALPHA)
    out.update({'gen_' + k: v for k, v in summarise(G2).items()})
    out['gen_losses'] = np.array([L2['D_KL'], L2['logP'], L2['W_dist'], L2['l_E'], L2['l_G']])
    return out


if __name__ == '__main__':
    np.savez_compressed(os.path.join(HERE, 'vawgan_F4_seed21.npz'), **run())
    print(os.path.getsize(os.path.join(HERE, 'vawgan_F4_seed21.npz')), 'bytes')
