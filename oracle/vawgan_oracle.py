"""CPU oracle of the VAWGAN branch -- TEST INFRASTRUCTURE, PARITY UNPINNED.

What the reference tree holds of this branch is the *trainer* (trainer/vae.py:115-218), the
architecture file (architecture-vawgan-vcc2016.json) and the layer helpers (util/layers.py); the model
class it trains lives on an un-vendored git branch (README.md:3).  This file is therefore a
SPECIFICATION, assembled from what the tree does pin down, and everything else is marked [spec]:

* the loss dictionary the trainer consumes: keys ``l_D, l_E, l_G, D_KL, logP, W_dist, gp``
  (trainer/vae.py:141-143,196-201);
* variable groups by name: 'Discriminator' / 'Encoder' / 'Generator' + 'y_emb' (trainer/vae.py:128-130);
* the discriminator's layer table: kernels [7,7,115] x 1, stride 3, outputs 16/32/64
  (architecture-vawgan-vcc2016.json:7-14), built [spec] with the tree's own conv helper
  ``conv2d_nchw_layernorm`` + ``lrelu`` (util/layers.py:47-66,147-149; the encoder's block) and one dense
  unit on the flattened last layer;
* hyper-parameters ``nIterD`` 5, ``lambda`` 10, ``alpha`` 50 (json:36-39);
* [spec] losses after the WGAN-GP formulation the trainer's docstring links to
  (trainer/vae.py:117-120, improved_wgan_training):
      W_dist = mean D(x) - mean D(xh)
      gp     = mean_f (|| d D(xi_f) / d xi_f ||_2 - 1)^2 ,  xi = x + u (xh - x),  u ~ U[0,1) per frame
      l_D = -W_dist + lambda gp ;  l_E = -logP + D_KL ;  l_G = -logP + alpha W_dist
  (``merge_dim``, ``feature_layer`` and ``l2-reg`` of the json are not used, like the keys the ConvVAE
  never reads);
* optimiser: ONE tf.train.AdamOptimizer shared by three minimize ops (trainer/vae.py:126,141-145):
  the Adam slots are per variable, the beta powers belong to the optimizer, so in TF 1.2 every apply
  -- 5 critic steps, the encoder step and the generator step of one iteration -- advances the same
  power accumulators: the t of the bias correction counts APPLIES, not iterations [TF1-semantics].

The gradient oracle is PyTorch float64 autograd (the gradient penalty needs the second derivative:
``create_graph=True``).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from . import convvae_oracle as O


def disc_geometry(arch):
    """TF 'SAME' shape chain of the discriminator convs (util/layers.py:56-64 with padding='SAME')."""
    d = arch['discriminator']
    assert len(d['output']) == len(d['kernel']) == len(d['stride'])
    c, h, out = 1, int(arch['hwc'][0]), []
    for o, k, s in zip(d['output'], d['kernel'], d['stride']):
        ho, plo, _ = O.same_pad_conv(h, int(k[0]), int(s[0]))
        out.append(dict(cin=c, hin=h, cout=int(o), hout=ho, k=int(k[0]), s=int(s[0]), pad=plo))
        c, h = int(o), ho
    return out


def disc_param_layout(arch):
    """name -> shape, creation order [spec; TF1 naming of the encoder block under the 'Discriminator' template]."""
    lay = OrderedDict()
    g = disc_geometry(arch)
    for i, l in enumerate(g):
        p = 'Discriminator/Conv2d-%d/' % i
        lay[p + 'kernel'] = (l['k'], 1, l['cin'], l['cout'])
        lay[p + 'bias'] = (l['cout'],)
        lay[p + 'layernorm.offset'] = (l['cout'], 1, 1)
        lay[p + 'layernorm.scale'] = (l['cout'], 1, 1)
    lay['Discriminator/dense/kernel'] = (g[-1]['cout'] * g[-1]['hout'], 1)
    lay['Discriminator/dense/bias'] = (1,)
    return lay


def disc_init_params(arch, seed=0, perturb_ln=True, bias_scale=0.05):
    """Glorot-uniform kernels; biases / LN parameters perturbed so that no gradient is trivially zero."""
    rng = np.random.RandomState(seed)
    D = OrderedDict()
    for name, shape in disc_param_layout(arch).items():
        n = int(np.prod(shape))
        if name.endswith('.scale'):
            v = 1.0 + (0.1 * rng.randn(n) if perturb_ln else 0.0) * np.ones(n)
        elif name.endswith('.offset') or name.endswith('bias'):
            v = bias_scale * rng.randn(n) if perturb_ln else np.zeros(n)
        else:
            if len(shape) == 4:
                fi, fo = shape[0] * shape[2], shape[0] * shape[3]
            else:
                fi, fo = shape
            lim = np.sqrt(6.0 / (fi + fo))
            v = rng.uniform(-lim, lim, n)
        D[name] = np.asarray(v, np.float64).reshape(shape)
    return D


def flatten(D):
    return np.concatenate([np.asarray(v, np.float64).reshape(-1) for v in D.values()])


def torch_discriminate(arch, D, x):
    """x [B, H] -> critic value d [B]."""
    B = x.shape[0]
    cur = x.reshape(B, 1, -1, 1)
    for i, l in enumerate(disc_geometry(arch)):
        p = 'Discriminator/Conv2d-%d/' % i
        a = O.torch_conv_same(cur, D[p + 'kernel'], D[p + 'bias'], l['s'])
        cur = O.torch_lrelu(O.torch_layernorm(a, D[p + 'layernorm.offset'], D[p + 'layernorm.scale']))
    return (cur.reshape(B, -1) @ D['Discriminator/dense/kernel']).reshape(B) + D['Discriminator/dense/bias']


def torch_critic_terms(arch, D, x, xh, u, create_graph):
    """W_dist, gp and the per-frame input gradient of the critic at the interpolates."""
    import torch
    xi = (x + u.reshape(-1, 1) * (xh - x)).detach().requires_grad_(True)
    d_real, d_fake, d_int = torch_discriminate(arch, D, x), torch_discriminate(arch, D, xh), torch_discriminate(arch, D, xi)
    g, = torch.autograd.grad(d_int.sum(), xi, create_graph=create_graph)
    norm = torch.sqrt((g ** 2).sum(-1))
    return dict(W_dist=d_real.mean() - d_fake.mean(), gp=((norm - 1.0) ** 2).mean(), g=g, norm=norm,
                d_real=d_real, d_fake=d_fake, d_int=d_int)


def critic_loss_and_grads(arch, D_np, x, xh, u, lam, dtype=None):
    """l_D = -W_dist + lam gp and its gradient w.r.t. every discriminator tensor."""
    import torch
    dtype = dtype or torch.float64
    D = O.torch_params(D_np, dtype, requires_grad=True)
    xt, xht, ut = (torch.tensor(np.asarray(a), dtype=dtype) for a in (x, xh, u))
    T = torch_critic_terms(arch, D, xt, xht, ut, create_graph=True)
    l_D = -T['W_dist'] + lam * T['gp']
    l_D.backward()
    grads = OrderedDict((k, v.grad.numpy().copy()) for k, v in D.items())
    out = {k: v.detach().numpy().copy() for k, v in T.items()}
    out["l_D"] = float(l_D.detach())
    return out, grads


def encoder_generator_grads(arch, P_np, D_np, x, y, eps, alpha, dtype=None):
    """The second half of an iteration (trainer/vae.py:141-145): l_E = -logP + D_KL differentiated w.r.t. the
    'Encoder' tensors, l_G = -logP + alpha W_dist w.r.t. 'Generator' + 'y_emb', both from ONE forward pass
    with the pre-update parameters.  Returns (losses, OrderedDict name -> grad over the ConvVAE table)."""
    import torch
    dtype = dtype or torch.float64
    P = O.torch_params(P_np, dtype, requires_grad=True)
    D = O.torch_params(D_np, dtype)
    xt = torch.tensor(np.asarray(x), dtype=dtype)
    yt = torch.tensor(np.asarray(y), dtype=torch.int64)
    et = torch.tensor(np.asarray(eps), dtype=dtype)
    L = O.torch_loss(arch, P, xt, yt, et)
    W = torch_discriminate(arch, D, xt).mean() - torch_discriminate(arch, D, L['xh']).mean()
    l_E = -L['logP'] + L['D_KL']
    l_G = -L['logP'] + alpha * W
    e_names = [k for k in P if 'Encoder' in k]
    g_names = [k for k in P if 'Generator' in k or 'y_emb' in k]
    ge = torch.autograd.grad(l_E, [P[k] for k in e_names], retain_graph=True)
    gg = torch.autograd.grad(l_G, [P[k] for k in g_names])
    grads = OrderedDict((k, None) for k in P)
    for k, g in list(zip(e_names, ge)) + list(zip(g_names, gg)):
        grads[k] = g.numpy().copy()
    losses = dict(D_KL=float(L['D_KL'].detach()), logP=float(L['logP'].detach()), W_dist=float(W.detach()),
                  l_E=float(l_E.detach()), l_G=float(l_G.detach()),
                  xh=L['xh'].detach().numpy().copy())
    return losses, grads


def forward_xh(arch, P_np, x, y, eps):
    """xh of the current ConvVAE parameters (float64)."""
    return O.np_forward(arch, P_np, x, y, eps)['xh']


def train_iterations(arch, P_np, D_np, batches, lr, b1, b2, alpha, lam, n_iter_d):
    """`len(batches) // (n_iter_d + 1)` iterations of trainer/vae.py:176-179.  Every sess.run dequeues its own
    batch: `batches` is the flat sequence of dicts(x, y, eps, u) in consumption order (u is unused by the
    generator step).  One Adam apply counter shared by the three groups (see the header).  Returns the final
    (P, D), the per-iteration losses of the generator step, and per tensor the mask of entries whose gradient was
    above 1e-2 of its tensor's largest in EVERY apply (Adam normalises each entry by its own history, so entries at
    a low-precision implementation's noise floor follow no particular trajectory)."""
    P = OrderedDict((k, np.array(v, np.float64)) for k, v in P_np.items())
    D = OrderedDict((k, np.array(v, np.float64)) for k, v in D_np.items())
    mP = {k: np.zeros_like(v) for k, v in P.items()}
    vP = {k: np.zeros_like(v) for k, v in P.items()}
    mD = {k: np.zeros_like(v) for k, v in D.items()}
    vD = {k: np.zeros_like(v) for k, v in D.items()}
    strong = {k: np.ones(v.shape, bool) for k, v in list(P.items()) + list(D.items())}

    def note(k, g):
        strong[k] &= np.abs(g) > 1e-2 * np.abs(g).max()
    t, log, it = 0, [], iter(batches)
    for _ in range(len(batches) // (n_iter_d + 1)):
        for _ in range(n_iter_d):                      # sess.run(self.opt['d']) x nIterD
            b = next(it)
            xh = forward_xh(arch, P, b['x'], b['y'], b['eps'])
            _, g = critic_loss_and_grads(arch, D, b['x'], xh, b['u'], lam)
            t += 1
            for k in D:
                note(k, g[k])
                D[k], mD[k], vD[k] = O.tf_adam_step(D[k], g[k], mD[k], vD[k], t, lr, b1, b2)
        b = next(it)                                    # sess.run(self.opt['g']): opt_e, then opt_g
        losses, g = encoder_generator_grads(arch, P, D, b['x'], b['y'], b['eps'], alpha)
        t += 1
        for k in P:
            if 'Encoder' in k:
                note(k, g[k])
                P[k], mP[k], vP[k] = O.tf_adam_step(P[k], g[k], mP[k], vP[k], t, lr, b1, b2)
        t += 1
        for k in P:
            if 'Generator' in k or 'y_emb' in k:
                note(k, g[k])
                P[k], mP[k], vP[k] = O.tf_adam_step(P[k], g[k], mP[k], vP[k], t, lr, b1, b2)
        losses.pop('xh')
        log.append(losses)
    return P, D, log, strong


# ---------------------------------------------------------------------------------------------------------
# closed forms the HIP kernels implement (checked against autograd in tests/test_vawgan_oracle.py)
# ---------------------------------------------------------------------------------------------------------
def np_ln_bwd(p, xhat, r):
    """LayerNorm input gradient for one frame: p = gamma * d(LN output), xhat normalised input, r = rstd."""
    return r * (p - p.mean() - xhat * (p * xhat).mean())


def np_ln_bwd_bwd(q, p, xhat, r):
    """Adjoint of np_ln_bwd for one frame: given q = adjoint of its result, returns (adjoint of p, adjoint of the
    layer's pre-LN input u).  The map p -> np_ln_bwd(p) is symmetric, so the first is the same operator applied
    to q; the second collects the dependence of xhat and r on u."""
    ub = np_ln_bwd(p, xhat, r)
    pt = np_ln_bwd(q, xhat, r)
    m2, mq = (p * xhat).mean(), (q * xhat).mean()
    xt = -r * (m2 * q + mq * p)                       # adjoint of xhat
    ut = r * (xt - xt.mean() - xhat * (q * ub + xt * xhat).mean())
    return pt, ut
