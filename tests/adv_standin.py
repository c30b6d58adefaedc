"""CPU stand-ins for hipvae.Engine / hipvae.critic.Critic built from the oracle (TEST INFRASTRUCTURE): they mirror
the contract hipvae.adversarial.AdvStepper relies on, so that its host logic -- variable groups, the shared Adam
apply counter, the shifted-target formulation of the generator gradient, the data-parallel reductions -- runs on
CPU (and under gloo) in float64."""
from collections import OrderedDict

import numpy as np
import torch

from helpers import SMALL_ARCH
from oracle import convvae_oracle as O
from oracle import vawgan_oracle as V
from oracle import philox_ref

SMALL_VAWGAN = dict(SMALL_ARCH)
SMALL_VAWGAN['discriminator'] = {"kernel": [[5, 1], [4, 1]], "stride": [[3, 1], [3, 1]], "output": [3, 4]}
SMALL_VAWGAN['training'] = dict(SMALL_ARCH['training'], nIterD=2, alpha=50.0)
SMALL_VAWGAN['training']['lambda'] = 10.0


def _layout(shapes):
    out, off = OrderedDict(), 0
    for n, shp in shapes.items():
        out[n] = (off, tuple(shp))
        off += int(np.prod(shp))
    return out, off


class EngineStandIn(object):
    def __init__(self, arch, seed):
        self.arch = arch
        self.layout, self.n_params = _layout(O.param_layout(arch))
        self.names = list(self.layout.keys())
        self.params = torch.tensor(O.flatten_params(O.init_params(arch, seed)), dtype=torch.float64)
        self._xh = None

    def _P(self):
        return O.unflatten_params(self.arch, self.params.numpy())

    def philox_normal(self, rows, seed, offset=0):
        z = self.arch['z_dim']
        return torch.tensor(philox_ref.normal(rows * z, seed, offset).reshape(rows, z), dtype=torch.float64)

    def philox_uniform(self, n, seed, offset=0):
        return torch.tensor(philox_ref.uniform(n, seed, offset), dtype=torch.float64)

    def ws_region(self, F, mode, name):
        assert name == 'xh'
        return self._xh.reshape(-1)

    def loss_fwd(self, x, y, eps, out=None):
        R = O.np_forward(self.arch, self._P(), x.numpy(), y.numpy(), eps.numpy())
        self._xh = torch.tensor(R['xh'])
        out.copy_(torch.tensor([R['G'], R['D_KL'], R['logP']]))
        return out

    def _step(self, x, y, eps, target, grads, out):
        P = O.torch_params(self._P(), torch.float64, requires_grad=True)
        xt = x.double()
        L = O.torch_loss(self.arch, P, xt, y, eps.double())
        tg = xt if target is None else target.double()
        lp = (-0.5 * (O.LOG_2PI + (tg.reshape(tg.shape[0], -1) - L['xh']) ** 2 / (1.0 + O.EPSILON))).sum(-1).mean()
        G = -lp + L['D_KL']
        G.backward()
        grads.copy_(torch.cat([(P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])).reshape(-1)
                               for n in self.names]))
        self._xh = L['xh'].detach().clone()
        out.copy_(torch.tensor([float(G.detach()), float(L['D_KL'].detach()), float(lp.detach())]))
        return out

    def train_fwd_bwd(self, x, y, eps, grads, out=None):
        return self._step(x, y, eps, None, grads, out)

    def train_fwd_bwd_target(self, x, y, eps, target, grads, out=None):
        return self._step(x, y, eps, target, grads, out)

    def adam_range(self, params, grads, m, v, lo, hi, step, lr, b1, b2, eps=1e-8, grad_scale=1.0):
        p, mm, vv = O.tf_adam_step(params[lo:hi].numpy(), grads[lo:hi].numpy() * grad_scale, m[lo:hi].numpy(),
                                   v[lo:hi].numpy(), step, lr, b1, b2, eps)
        params[lo:hi] = torch.tensor(p)
        m[lo:hi] = torch.tensor(mm)
        v[lo:hi] = torch.tensor(vv)


class CriticStandIn(object):
    def __init__(self, arch, seed):
        self.arch = arch
        self.layout, self.n_params = _layout(V.disc_param_layout(arch))
        self.names = list(self.layout.keys())
        self.params = torch.tensor(V.flatten(V.disc_init_params(arch, seed)), dtype=torch.float64)

    def _D(self):
        flat, out = self.params.numpy(), OrderedDict()
        for n, (off, shp) in self.layout.items():
            out[n] = flat[off:off + int(np.prod(shp))].reshape(shp).copy()
        return out

    def critic_fwd_bwd(self, x, xh, t, lam, grads, out=None):
        F = x.shape[0]
        L, G = V.critic_loss_and_grads(self.arch, self._D(), x.numpy().reshape(F, -1), xh.numpy().reshape(F, -1),
                                       t.numpy(), lam)
        grads.copy_(torch.tensor(np.concatenate([G[n].ravel() for n in self.names])))
        out.copy_(torch.tensor([float(L['W_dist']), float(L['gp'])]))
        return out

    def generator_target(self, x, xh, alpha, out=None):
        F = x.shape[0]
        D = O.torch_params(self._D(), torch.float64)
        xt = x.double().reshape(F, -1)
        xht = xh.double().reshape(F, -1).clone().requires_grad_(True)
        d_fake = V.torch_discriminate(self.arch, D, xht)
        g, = torch.autograd.grad(d_fake.sum(), xht)
        W = V.torch_discriminate(self.arch, D, xt).mean() - d_fake.mean()
        out.copy_(torch.tensor([float(W.detach()), 0.0]))
        return xt + alpha * (1.0 + O.EPSILON) * g, out


def adv_batches(arch, F, n, seed):
    out = []
    for i in range(n):
        x, y, eps = O.make_inputs(arch, F, seed + i)
        x[F // 2:] *= 0.25                  # shards with visibly different content
        out.append(dict(x=x, y=y, eps=eps, u=np.random.RandomState(seed + 1000 + i).rand(F)))
    return out
