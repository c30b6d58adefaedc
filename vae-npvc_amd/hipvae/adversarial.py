"""The three `minimize` ops of the VAWGAN trainer (trainer/vae.py:115-147) as host logic over the C-ABI.

    opt['d']  critic step      : l_D = -W_dist + lambda gp         w.r.t. the 'Discriminator' tensors
    opt['g']  generator step   : opt_e  l_E = -logP + D_KL         w.r.t. 'Encoder'
                                 opt_g  l_G = -logP + alpha W_dist w.r.t. 'Generator' + 'y_emb'   (after opt_e)

One tf.train.AdamOptimizer serves all three (trainer/vae.py:126): per-variable slots, ONE pair of beta-power
accumulators, so the t of Adam's bias correction counts applies (TF 1.2 semantics; `applies` below).
`global_step` counts generator steps only (trainer/vae.py:145).

The gradient of l_G reuses the ConvVAE backward unchanged: the adversarial term enters through the
reconstruction gradient, d l_G / d xh = (xh - x) / ((1 + 1e-6) F) - (alpha / F) dD(xh)/dxh, which is what the
backward computes for the shifted target x' = x + alpha (1 + 1e-6) dD(xh)/dxh
(vaenpvc_disc_generator_target + vaenpvc_train_bwd_target).  The generator step is therefore one ConvVAE
step (gradient of l_E), the critic's forward + input gradient, and one more ConvVAE BACKWARD pass on the same
activations (gradient of l_G).

Data parallel: frames are independent in every term (the penalty is per frame), so ranks shard frames and
all-reduce (SUM) each gradient buffer before its apply; 1/world goes into Adam's grad_scale.
"""
import os

import torch
import torch.distributed as dist

from . import lib as L
from .dp import world_info, rank_seed


def name_ranges(layout, pred):
    """Contiguous [lo, hi) float ranges covering the tensors whose name satisfies `pred` (table order)."""
    out = []
    for name, (off, shape) in layout.items():
        if not pred(name):
            continue
        n = 1
        for s in shape:
            n *= s
        if out and out[-1][1] == off:
            out[-1][1] = off + n
        else:
            out.append([off, off + n])
    return [tuple(r) for r in out]


class AdvStepper(object):
    def __init__(self, engine, critic, lr, beta1, beta2, alpha, lam, eps=1e-8, group=None, seed=0):
        self.backend, self.critic = engine, critic
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(beta1), float(beta2), float(eps)
        self.alpha, self.lam = float(alpha), float(lam)
        self.group = group
        self.rank, self.world = world_info(group)
        # VAENPVC_FORCE_DIST=1: run the collectives even with one rank (smoke-tests the RCCL path, as hipvae.dp)
        self.collective = self.world > 1 or (os.environ.get('VAENPVC_FORCE_DIST') == '1' and dist.is_available()
                                             and dist.is_initialized())
        p, d = engine.params, critic.params
        # Gradient buffers with the step's loss scalars in their tails (as hipvae.dp.Stepper): ONE all-reduce per
        # generator step (both ConvVAE gradients + {G, D_KL, logP} + W_dist) and ONE per critic step (gradient +
        # {W_dist, gp}); no collective of its own for any loss, so an exception between collectives cannot leave the
        # ranks with different numbers of pending all-reduces (round-2 advisor finding).
        al = lambda k: (k + 63) // 64 * 64
        n, nd = p.numel(), d.numel()
        self._gbuf = torch.zeros(2 * al(n) + 64, dtype=p.dtype, device=p.device)
        self.g_e = self._gbuf[:n]               # gradient of l_E (all tensors; the 'Encoder' ranges are applied)
        self.g_g = self._gbuf[al(n):al(n) + n]  # gradient of l_G (the 'Generator' / 'y_emb' ranges are applied)
        self._gtail = self._gbuf[2 * al(n):2 * al(n) + 8]
        self._dbuf = torch.zeros(al(nd) + 64, dtype=d.dtype, device=d.device)
        self.g_d = self._dbuf[:nd]
        self._dtail = self._dbuf[al(nd):al(nd) + 4]
        self.m, self.v = torch.zeros_like(p), torch.zeros_like(p)
        self.m_d, self.v_d = torch.zeros_like(d), torch.zeros_like(d)
        self.enc_ranges = name_ranges(engine.layout, lambda n: 'Encoder' in n)
        self.gen_ranges = name_ranges(engine.layout, lambda n: 'Generator' in n or 'y_emb' in n)
        self.applies = 0                        # Adam t: one per minimize op run
        self.step_count = 0                     # global_step
        self.seed = rank_seed(seed, self.rank, self.world)
        self._draws = 0
        # last values of the status line (trainer/vae.py:196-201); device tensors
        self.status = {k: torch.zeros((), device=p.device) for k in ('D_KL', 'logP', 'W_dist', 'gp')}
        self._l3 = self._gtail[0:3]             # {G, D_KL, logP} of the generator step's first pass
        self._l2 = self._gtail[4:6]             # {W_dist, -} of the generator step (critic forward on xh)
        self._l2d = self._dtail[0:2]            # {W_dist, gp} of a critic step
        self._l3b = torch.zeros(3, dtype=p.dtype, device=p.device)   # losses against the shifted target (not reported)
        self._l3f = torch.zeros(3, dtype=p.dtype, device=p.device)   # forward-only pass of the critic steps

    def broadcast_params(self, src=0):
        if self.world > 1:
            dist.broadcast(self.backend.params, src=src, group=self.group)
            dist.broadcast(self.critic.params, src=src, group=self.group)

    def _draw(self, F):
        """Sampler noise [F, z] and interpolation coefficients [F] of one sess.run (fresh per call, per rank)."""
        self._draws += 1
        be = self.backend
        return be.philox_normal(F, self.seed, self._draws), be.philox_uniform(F, self.seed ^ 0x7157, self._draws)

    def _reduce(self, buf):
        if self.collective:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    def _mean(self, t):
        """A loss vector from a gradient buffer's tail averaged over ranks (new tensor; the SUM arrived with the
        gradient all-reduce -- no collective here)."""
        t = t.clone()
        if self.collective:
            t /= float(self.world)
        return t

    # ------------------------------------------------------------------ sess.run(opt['d'])
    def critic_step(self, x, y, eps=None, t=None):
        return self.critic_steps([(x, y)], None if eps is None else [eps], None if t is None else [t])

    def critic_steps(self, batches, eps=None, t=None):
        """len(batches) consecutive `sess.run(opt['d'])` (trainer/vae.py:177-178), each on its own batch (x, y).  The
        encoder and the generator do not change between critic steps, so their forward pass runs ONCE over the
        concatenated batches (same kernels, more frames each) and every critic step takes its rows of xh: at 16
        frames per batch the device time is the number of kernels, and the generator forward was 60 of a critic
        step's ~130.  eps / t: optional lists of injected draws, one per batch."""
        be, cr = self.backend, self.critic
        n, F = len(batches), batches[0][0].shape[0]
        if any(b[0].shape[0] != F for b in batches):
            raise ValueError('critic batches must have one size')
        X = batches[0][0].reshape(F, -1) if n == 1 else torch.cat([b[0].reshape(F, -1) for b in batches])
        Y = batches[0][1].reshape(-1) if n == 1 else torch.cat([b[1].reshape(-1) for b in batches])
        if eps is None or t is None:
            e2, t2 = self._draw(n * F)
        eps = e2 if eps is None else (eps[0] if n == 1 else torch.cat(list(eps)))
        t = t2 if t is None else (t[0] if n == 1 else torch.cat(list(t)))
        be.loss_fwd(X, Y, eps, out=self._l3f)                        # forward only: xh of the current generator
        xh = be.ws_region(n * F, L.MODE_INFER, 'xh').view(n * F, -1)
        for i in range(n):
            lo, hi = i * F, (i + 1) * F
            cr.critic_fwd_bwd(X[lo:hi], xh[lo:hi], t[lo:hi], self.lam, self.g_d, out=self._l2d)
            self._reduce(self._dbuf)                                 # gradient + {W_dist, gp}: one collective
            self.applies += 1
            be.adam_range(cr.params, self.g_d, self.m_d, self.v_d, 0, cr.n_params, self.applies, self.lr, self.beta1,
                          self.beta2, self.eps, 1.0 / self.world)
        l2 = self._mean(self._l2d)
        self.status['W_dist'], self.status['gp'] = l2[0], l2[1]
        return l2

    # ------------------------------------------------------------------ sess.run(opt['g'])
    def generator_step(self, x, y, eps=None):
        be, cr = self.backend, self.critic
        F = x.shape[0]
        if eps is None:
            eps, _ = self._draw(F)
        be.train_fwd_bwd(x, y, eps, self.g_e, out=self._l3)          # l_E = G of the ConvVAE
        xh = be.ws_region(F, L.MODE_TRAIN, 'xh').view(F, -1)
        target, _ = cr.generator_target(x, xh, self.alpha, out=self._l2)
        # second gradient on the activations of the first pass (a backend without the backward-only entry reruns the step)
        getattr(be, 'train_bwd_target', be.train_fwd_bwd_target)(x, y, eps, target, self.g_g, out=self._l3b)
        self._reduce(self._gbuf)                                     # both gradients and the loss scalars: one collective
        gs = 1.0 / self.world
        self.applies += 1                                            # opt_e
        for lo, hi in self.enc_ranges:
            be.adam_range(be.params, self.g_e, self.m, self.v, lo, hi, self.applies, self.lr, self.beta1, self.beta2,
                          self.eps, gs)
        self.applies += 1                                            # opt_g (control-dependent on opt_e)
        for lo, hi in self.gen_ranges:
            be.adam_range(be.params, self.g_g, self.m, self.v, lo, hi, self.applies, self.lr, self.beta1, self.beta2,
                          self.eps, gs)
        self.step_count += 1
        l3, l2 = self._mean(self._l3), self._mean(self._l2)
        self.status['D_KL'], self.status['logP'], self.status['W_dist'] = l3[1], l3[2], l2[0]
        return {'D_KL': l3[1], 'logP': l3[2], 'W_dist': l2[0]}

    # ------------------------------------------------------------------ checkpoint
    def state_dict(self):
        return {'params': self.backend.params.detach().cpu(), 'm': self.m.cpu(), 'v': self.v.cpu(),
                'd_params': self.critic.params.detach().cpu(), 'd_m': self.m_d.cpu(), 'd_v': self.v_d.cpu(),
                'step': self.step_count, 'applies': self.applies, 'draws': self._draws}

    def load_state_dict(self, sd):
        dev = self.backend.params.device
        self.backend.params.copy_(sd['params'].to(dev))
        self.m.copy_(sd['m'].to(dev))
        self.v.copy_(sd['v'].to(dev))
        self.critic.params.copy_(sd['d_params'].to(dev))
        self.m_d.copy_(sd['d_m'].to(dev))
        self.v_d.copy_(sd['d_v'].to(dev))
        self.step_count, self.applies = int(sd['step']), int(sd['applies'])
        self._draws = int(sd.get('draws', 0))       # the sampler / interpolation streams continue where they stopped
