"""CPU oracle: restatement of the reference ConvVAE hot path (TEST INFRASTRUCTURE).

PARITY UNPINNED -- see ``oracle/__init__.py``.  Every function cites the
reference lines (paths relative to /root/reference) it restates.

Two independent implementations live here:

* ``np_*``    float64 NumPy, written straight from the definitions
              (explicit tap loops, no library convolution);
* ``torch_*`` PyTorch-CPU (``F.conv2d`` / ``F.conv_transpose2d`` + autograd),
              any dtype; its autograd is the gradient oracle.

They must agree to ~1e-12 in float64 before either is trusted
(tests/test_oracle.py).
"""
from __future__ import annotations

import json
import math
from collections import OrderedDict

import numpy as np

LN_EPS = 1e-5          # util/layers.py:44  (tf.nn.batch_normalization eps)
LRELU_LEAK = 0.02      # util/layers.py:147
EPSILON = 1e-6         # util/layers.py:7
LOG_2PI = math.log(2.0 * math.pi)   # util/layers.py:161


# --------------------------------------------------------------------------
# geometry  (architecture-vae-vcc2016.json; TF SAME rules)
# --------------------------------------------------------------------------
def load_arch(path):
    with open(path) as fp:
        return json.load(fp)


def same_pad_conv(h, k, s):
    """TF 'SAME' for conv: returns (h_out, pad_lo, pad_hi)."""
    h_out = -(-h // s)
    total = max((h_out - 1) * s + k - h, 0)
    return h_out, total // 2, total - total // 2


def same_pad_convT(h, k, s):
    """TF 'SAME' for conv2d_transpose (= input-gradient of the SAME conv that
    maps h*s -> h): returns (h_out, pad_lo)."""
    h_out = h * s
    total = max((h - 1) * s + k - h_out, 0)
    return h_out, total // 2


def geometry(arch):
    """Shape chain of model/vae.py:72-103 for an architecture dict."""
    enc, gen = arch['encoder'], arch['generator']
    # model/vae.py:37-39 sanity check
    for net in (enc, gen):
        assert len(net['output']) == len(net['kernel']) == len(net['stride'])
    H = arch['hwc'][0]
    g = {'enc': [], 'dec': [], 'z_dim': arch['z_dim'], 'y_dim': arch['y_dim'], 'H': H}
    c, h = 1, H
    for o, k, s in zip(enc['output'], enc['kernel'], enc['stride']):
        ho, plo, phi = same_pad_conv(h, k[0], s[0])
        g['enc'].append(dict(cin=c, hin=h, cout=o, hout=ho, k=k[0], s=s[0], pad=plo, pad_hi=phi))
        c, h = o, ho
    g['flat'] = c * h
    gh, gw, gc = gen['hwc']            # model/vae.py:86  (h, w, c)
    g['merge'] = gh * gw * gc
    c, h = gc, gh
    for o, k, s in zip(gen['output'], gen['kernel'], gen['stride']):
        ho, plo = same_pad_convT(h, k[0], s[0])
        g['dec'].append(dict(cin=c, hin=h, cout=o, hout=ho, k=k[0], s=s[0], pad=plo))
        c, h = o, ho
    g['out'] = c * h
    return g


def param_layout(arch):
    """The 44 trainables in TF creation order (y_emb in __init__, then Encoder,
    then Generator template bodies; model/vae.py:20-24,72-103; util/layers.py:33-64).
    Shapes are the TF shapes.  Returns OrderedDict name -> shape."""
    g = geometry(arch)
    L = OrderedDict()
    z = g['z_dim']
    L['y_embedding/y_emb'] = (g['y_dim'], z)            # width = z_dim (trap T3)
    for i, l in enumerate(g['enc']):
        p = 'Encoder/Conv2d-%d/' % i
        L[p + 'kernel'] = (l['k'], 1, l['cin'], l['cout'])
        L[p + 'bias'] = (l['cout'],)
        L[p + 'layernorm.offset'] = (l['cout'], 1, 1)
        L[p + 'layernorm.scale'] = (l['cout'], 1, 1)
    L['Encoder/dense/kernel'] = (g['flat'], z)
    L['Encoder/dense/bias'] = (z,)
    L['Encoder/dense_1/kernel'] = (g['flat'], z)
    L['Encoder/dense_1/bias'] = (z,)
    L['Generator/fully_connected/weights'] = (z, g['merge'])
    L['Generator/fully_connected/biases'] = (g['merge'],)
    L['Generator/fully_connected_1/weights'] = (z, g['merge'])
    L['Generator/fully_connected_1/biases'] = (g['merge'],)
    L['Generator/BiasAdd/biases'] = (g['merge'],)
    nd = len(g['dec'])
    for i, l in enumerate(g['dec']):
        p = 'Generator/conv2d_transpose%s/' % ('' if i == 0 else '_%d' % i)
        L[p + 'kernel'] = (l['k'], 1, l['cout'], l['cin'])
        L[p + 'bias'] = (l['cout'],)
        if i < nd - 1:
            L['Generator/ConvT-LN%d.offset' % i] = (l['cout'], 1, 1)
            L['Generator/ConvT-LN%d.scale' % i] = (l['cout'], 1, 1)
    return L


def init_params(arch, seed=0, perturb_ln=True, bias_scale=0.05):
    """Seeded parameters.  Weights: Glorot-uniform (TF default for get_variable /
    tf.layers / slim: limit sqrt(6/(fan_in+fan_out)), conv fans = kh*kw*shape[-2],
    kh*kw*shape[-1]).  The reference zero-inits biases / LN offset and one-inits
    LN scale; for parity tests we perturb them (perturb_ln, bias_scale) so that
    every gradient path carries signal."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = OrderedDict()
    for name, shp in param_layout(arch).items():
        if name.endswith('.scale'):
            v = np.ones(shp) + (0.1 * rng.uniform(-1, 1, shp) if perturb_ln else 0.0)
        elif name.endswith('.offset'):
            v = 0.1 * rng.uniform(-1, 1, shp) if perturb_ln else np.zeros(shp)
        elif name.endswith('bias') or name.endswith('biases'):
            v = bias_scale * rng.uniform(-1, 1, shp)
        else:
            if len(shp) == 4:
                rf = shp[0] * shp[1]
                fan_in, fan_out = rf * shp[2], rf * shp[3]
            else:
                fan_in, fan_out = shp[0], shp[1]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            v = rng.uniform(-lim, lim, shp)
        P[name] = v.astype(np.float32)
    return P


def make_inputs(arch, F, seed=0):
    """Synthetic x~U(-1,1) f32 [F,H], y int64 [F], eps~N(0,1) f32 [F,z]."""
    rng = np.random.Generator(np.random.PCG64(seed + 1000003))
    H = arch['hwc'][0]
    x = rng.uniform(-1, 1, (F, H)).astype(np.float32)
    y = rng.integers(0, arch['y_dim'], (F,)).astype(np.int64)
    eps = rng.standard_normal((F, arch['z_dim'])).astype(np.float32)
    return x, y, eps


def flatten_params(P):
    return np.concatenate([np.asarray(v, np.float32).ravel() for v in P.values()])


def unflatten_params(arch, flat):
    P, o = OrderedDict(), 0
    for name, shp in param_layout(arch).items():
        n = int(np.prod(shp))
        P[name] = np.asarray(flat[o:o + n]).reshape(shp)
        o += n
    assert o == len(flat)
    return P


# --------------------------------------------------------------------------
# float64 NumPy, direct definitions
# --------------------------------------------------------------------------
def np_lrelu(x):                       # util/layers.py:147-149
    return np.maximum(x, LRELU_LEAK * x)


def np_layernorm(a, offset, scale):    # util/layers.py:10-44 ; a [F,C,H]
    mu = a.mean(axis=(1, 2), keepdims=True)
    var = ((a - mu) ** 2).mean(axis=(1, 2), keepdims=True)   # biased (tf.nn.moments)
    g = np.asarray(scale, a.dtype).reshape(1, -1, 1)
    b = np.asarray(offset, a.dtype).reshape(1, -1, 1)
    return (a - mu) / np.sqrt(var + LN_EPS) * g + b


def np_conv_same(x, W, b, s):
    """tf.layers.conv2d channels_first SAME, W TF layout [k,1,Cin,Cout]
    (util/layers.py:56-64); x [F,Cin,H] -> [F,Cout,Hout]."""
    F, cin, H = x.shape
    k = W.shape[0]
    ho, plo, phi = same_pad_conv(H, k, s)
    xp = np.zeros((F, cin, H + plo + phi), x.dtype)
    xp[:, :, plo:plo + H] = x
    out = np.zeros((F, W.shape[3], ho), x.dtype)
    for t in range(k):
        out += np.einsum('fcj,co->foj', xp[:, :, t:t + s * (ho - 1) + 1:s], W[t, 0])
    return out + np.asarray(b, x.dtype).reshape(1, -1, 1)


def np_convT_same(x, W, b, s):
    """tf.layers.conv2d_transpose channels_first SAME, W TF layout [k,1,Cout,Cin]
    (model/vae.py:96-99): out[o,p] = b[o] + sum_c sum_j W[p+pad-j*s, o, c] in[c,j]."""
    F, cin, H = x.shape
    k, cout = W.shape[0], W.shape[2]
    ho, pad = same_pad_convT(H, k, s)
    out = np.zeros((F, cout, ho), x.dtype)
    j = np.arange(H)
    for t in range(k):
        p = s * j - pad + t
        ok = (p >= 0) & (p < ho)
        if not ok.any():
            continue
        out[:, :, p[ok]] += np.einsum('fcj,oc->foj', x[:, :, ok], W[t, 0])
    return out + np.asarray(b, x.dtype).reshape(1, -1, 1)


def np_forward(arch, P, x, y, eps=None, dtype=np.float64):
    """Full forward of model/vae.py:106-137 (eps given) or the convert.py:85-89
    inference wiring z = z_mu (eps None).  Returns a dict with every
    intermediate (pre-LN conv outputs 'enc_a%d'/'dec_a%d', post-activation
    'enc_y%d'/'dec_y%d', z_mu, z_lv, z, h, xh, D_KL, logP, G)."""
    g = geometry(arch)
    Pd = {k: np.asarray(v, dtype) for k, v in P.items()}
    names = list(param_layout(arch).keys())
    R = {}
    F = x.shape[0]
    cur = np.asarray(x, dtype).reshape(F, 1, g['H'])
    for i, l in enumerate(g['enc']):                          # model/vae.py:72-78
        p = 'Encoder/Conv2d-%d/' % i
        a = np_conv_same(cur, Pd[p + 'kernel'], Pd[p + 'bias'], l['s'])
        cur = np_lrelu(np_layernorm(a, Pd[p + 'layernorm.offset'], Pd[p + 'layernorm.scale']))
        R['enc_a%d' % i], R['enc_y%d' % i] = a, cur
    flat = cur.reshape(F, -1)                                 # slim.flatten, C-major
    z_mu = flat @ Pd['Encoder/dense/kernel'] + Pd['Encoder/dense/bias']        # vae.py:80
    z_lv = flat @ Pd['Encoder/dense_1/kernel'] + Pd['Encoder/dense_1/bias']    # vae.py:81
    R['z_mu'], R['z_lv'] = z_mu, z_lv
    if eps is None:
        z = z_mu                                              # vae.py:139-141
    else:
        z = z_mu + np.asarray(eps, dtype) * np.sqrt(np.exp(z_lv))   # layers.py:152-156
    R['z'] = z
    R['xh'] = np_decode(arch, Pd, z, y, dtype, R)
    xf = np.asarray(x, dtype).reshape(F, -1)
    # layers.py:170-183 with mu2 = lv2 = 0 ; vae.py:112-119
    kld = 0.5 * ((0.0 - z_lv) + (np.exp(z_lv) + z_mu ** 2) / (1.0 + EPSILON) - 1.0)
    R['D_KL'] = kld.sum(-1).mean()
    # layers.py:159-167 with log_var = 0 ; vae.py:120-125
    lp = -0.5 * (LOG_2PI + 0.0 + (xf - R['xh']) ** 2 / (1.0 + EPSILON))
    R['logP'] = lp.sum(-1).mean()
    R['G'] = -R['logP'] + R['D_KL']                           # vae.py:128
    return R


def np_decode(arch, Pd, z, y, dtype=np.float64, R=None):
    """model/vae.py:84-103 generator; returns xh [F, H] (NHWC transpose of
    util/image.py:4-5 is a no-op on memory because W = 1)."""
    g = geometry(arch)
    Pd = {k: np.asarray(v, dtype) for k, v in Pd.items()}
    z = np.asarray(z, dtype)
    F = z.shape[0]
    e = Pd['y_embedding/y_emb'][np.asarray(y, np.int64)]      # vae.py:89
    h = (z @ Pd['Generator/fully_connected/weights'] + Pd['Generator/fully_connected/biases']
         + e @ Pd['Generator/fully_connected_1/weights'] + Pd['Generator/fully_connected_1/biases']
         + Pd['Generator/BiasAdd/biases'])                    # vae.py:51-61
    cur = h.reshape(F, g['dec'][0]['cin'], g['dec'][0]['hin'])   # vae.py:94
    if R is not None:
        R['h'] = h
    nd = len(g['dec'])
    for i, l in enumerate(g['dec']):
        p = 'Generator/conv2d_transpose%s/' % ('' if i == 0 else '_%d' % i)
        a = np_convT_same(cur, Pd[p + 'kernel'], Pd[p + 'bias'], l['s'])
        if i < nd - 1:                                        # vae.py:100-102
            cur = np_lrelu(np_layernorm(a, Pd['Generator/ConvT-LN%d.offset' % i],
                                        Pd['Generator/ConvT-LN%d.scale' % i]))
            if R is not None:
                R['dec_a%d' % i], R['dec_y%d' % i] = a, cur
        else:
            cur = a
    return cur.reshape(F, -1)


# --------------------------------------------------------------------------
# PyTorch-CPU restatement (+ autograd = gradient oracle)
# --------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def torch_params(P, dtype=None, requires_grad=False):
    torch = _torch()
    dtype = dtype or torch.float32
    out = OrderedDict()
    for k, v in P.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def torch_layernorm(a, offset, scale, want_xhat=False):
    torch = _torch()
    mu = a.mean(dim=(1, 2, 3), keepdim=True)
    var = ((a - mu) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    xhat = (a - mu) * torch.rsqrt(var + LN_EPS)
    n = xhat * scale.reshape(1, -1, 1, 1) + offset.reshape(1, -1, 1, 1)
    return (n, xhat) if want_xhat else n


def torch_lrelu(x, branch=None, tau=0.0):
    """max(x, 0.02 x) (util/layers.py:147-149).  lrelu is not differentiable at 0: an evaluation in another precision can
    land on the other side of the kink for |x| ~ its rounding error and then takes slope 0.02 instead of 1 (or vice
    versa) -- a finite, legitimate difference in that frame's gradient.  `branch` (bool tensor, True = positive side)
    lets a comparison pin the branch of the units with |x| < tau to the one the implementation under test took, so
    that what is compared is arithmetic, not which side of a kink a rounding error fell on."""
    torch = _torch()
    if branch is None:
        return torch.maximum(x, LRELU_LEAK * x)
    pos = torch.where(x.abs() < tau, branch, x >= 0)
    return torch.where(pos, x, LRELU_LEAK * x)


def torch_conv_same(x, W, b, s):
    import torch.nn.functional as Fn
    H, k = x.shape[2], W.shape[0]
    _, plo, phi = same_pad_conv(H, k, s)
    xp = Fn.pad(x, (0, 0, plo, phi))
    return Fn.conv2d(xp, W.permute(3, 2, 0, 1), b, stride=(s, 1))


def torch_convT_same(x, W, b, s):
    import torch.nn.functional as Fn
    H, k = x.shape[2], W.shape[0]
    ho, pad = same_pad_convT(H, k, s)
    full = Fn.conv_transpose2d(x, W.permute(3, 2, 0, 1), None, stride=(s, 1))
    out = full[:, :, pad:pad + ho, :]
    return out if b is None else out + b.reshape(1, -1, 1, 1)


def _tape(tape, name, t, factor=None):
    """tape: optional dict; records, under the NAME of a parameter whose gradient is a plain sum over the broadcast axes -- an additive
    parameter (conv / dense bias, LayerNorm offset: gradient = sum of the upstream gradient d) or a LayerNorm scale (gradient = sum of
    d * xhat: `factor` = xhat) -- the tensor it acts on, with its gradient retained (see torch_loss_and_grads(sum_scales=True))"""
    if tape is not None:
        t.retain_grad()
        tape[name] = (t, None if factor is None else factor.detach())


def _tape_lin(tape, name, fn, x, out):
    """... and for a WEIGHT tensor W of a linear layer out = fn(x, W) (+ bias): its gradient is sum over (frame, position) of x * d, so
    S = fn's weight gradient evaluated on |x| and |d| (the layer is linear in W: autograd of fn(|x|, W) with upstream |d|)"""
    if tape is not None:
        out.retain_grad()
        tape.setdefault('__lin__', []).append((name, fn, x.detach(), out))


def torch_encode(arch, P, x, kink=None, tape=None):
    """kink: optional {'tau': t, 'enc<i>': bool [F,C,H,1], 'dec<i>': ...} branch pins for torch_lrelu"""
    g = geometry(arch)
    F = x.shape[0]
    cur = x.reshape(F, 1, g['H'], 1)
    acts = []
    for i, l in enumerate(g['enc']):
        p = 'Encoder/Conv2d-%d/' % i
        a = torch_conv_same(cur, P[p + 'kernel'], P[p + 'bias'], l['s'])
        n, xhat = torch_layernorm(a, P[p + 'layernorm.offset'], P[p + 'layernorm.scale'], want_xhat=True)
        _tape(tape, p + 'bias', a)
        _tape_lin(tape, p + 'kernel', (lambda xx, W, s_=l['s']: torch_conv_same(xx, W, None, s_)), cur, a)
        _tape(tape, p + 'layernorm.offset', n)
        _tape(tape, p + 'layernorm.scale', n, xhat)
        cur = torch_lrelu(n) if kink is None else torch_lrelu(n, kink['enc%d' % i], kink['tau'])
        acts.append((a, cur))
    flat = cur.reshape(F, -1)
    z_mu = flat @ P['Encoder/dense/kernel'] + P['Encoder/dense/bias']
    z_lv = flat @ P['Encoder/dense_1/kernel'] + P['Encoder/dense_1/bias']
    _tape(tape, 'Encoder/dense/bias', z_mu)
    _tape(tape, 'Encoder/dense_1/bias', z_lv)
    _tape_lin(tape, 'Encoder/dense/kernel', (lambda xx, W: xx @ W), flat, z_mu)
    _tape_lin(tape, 'Encoder/dense_1/kernel', (lambda xx, W: xx @ W), flat, z_lv)
    return z_mu, z_lv, acts


def torch_decode(arch, P, z, y, kink=None, tape=None):
    torch = _torch()
    g = geometry(arch)
    F = z.shape[0]
    e = P['y_embedding/y_emb'][y]
    h = (z @ P['Generator/fully_connected/weights'] + P['Generator/fully_connected/biases']
         + e @ P['Generator/fully_connected_1/weights'] + P['Generator/fully_connected_1/biases']
         + P['Generator/BiasAdd/biases'])
    for nm in ('Generator/fully_connected/biases', 'Generator/fully_connected_1/biases', 'Generator/BiasAdd/biases'):
        _tape(tape, nm, h)
    _tape_lin(tape, 'Generator/fully_connected/weights', (lambda xx, W: xx @ W), z, h)
    _tape_lin(tape, 'Generator/fully_connected_1/weights', (lambda xx, W: xx @ W), e, h)
    if tape is not None:   # the embedding table: e = onehot(y) E, linear in E
        onehot = torch.zeros(F, P['y_embedding/y_emb'].shape[0], dtype=z.dtype)
        onehot[torch.arange(F), y] = 1.0
        _tape_lin(tape, 'y_embedding/y_emb', (lambda xx, W: xx @ W), onehot, e)
    cur = h.reshape(F, g['dec'][0]['cin'], g['dec'][0]['hin'], 1)
    nd = len(g['dec'])
    acts = [h]
    for i, l in enumerate(g['dec']):
        p = 'Generator/conv2d_transpose%s/' % ('' if i == 0 else '_%d' % i)
        a = torch_convT_same(cur, P[p + 'kernel'], P[p + 'bias'], l['s'])
        _tape(tape, p + 'bias', a)
        _tape_lin(tape, p + 'kernel', (lambda xx, W, s_=l['s']: torch_convT_same(xx, W, None, s_)), cur, a)
        if i < nd - 1:
            n, xhat = torch_layernorm(a, P['Generator/ConvT-LN%d.offset' % i], P['Generator/ConvT-LN%d.scale' % i], want_xhat=True)
            _tape(tape, 'Generator/ConvT-LN%d.offset' % i, n)
            _tape(tape, 'Generator/ConvT-LN%d.scale' % i, n, xhat)
            cur = torch_lrelu(n) if kink is None else torch_lrelu(n, kink['dec%d' % i], kink['tau'])
            acts.append(a)
        else:
            cur = a
    return cur.reshape(F, -1), acts


def torch_loss(arch, P, x, y, eps, kink=None, tape=None):
    """model/vae.py:106-137 -> dict(G, D_KL, logP, z_mu, z_lv, xh)."""
    torch = _torch()
    z_mu, z_lv, _ = torch_encode(arch, P, x, kink, tape)
    z = z_mu + eps * torch.sqrt(torch.exp(z_lv))
    xh, _ = torch_decode(arch, P, z, y, kink, tape)
    kld = 0.5 * ((0.0 - z_lv) + (torch.exp(z_lv) + z_mu ** 2) / (1.0 + EPSILON) - 1.0)
    D_KL = kld.sum(-1).mean()
    lp = -0.5 * (LOG_2PI + (x.reshape(x.shape[0], -1) - xh) ** 2 / (1.0 + EPSILON))
    logP = lp.sum(-1).mean()
    return dict(G=-logP + D_KL, D_KL=D_KL, logP=logP, z_mu=z_mu, z_lv=z_lv, xh=xh, z=z)


def torch_loss_and_grads(arch, P_np, x, y, eps, dtype=None, kink=None, sum_scales=False):
    """Returns (losses dict of floats/arrays, OrderedDict name -> grad ndarray).  kink: see torch_encode / torch_lrelu
    (branch pins as bool ndarrays [F,C,H], plus 'tau').

    sum_scales=True returns a third value: for every ADDITIVE parameter b (conv / dense biases, LayerNorm offsets, the three
    merge biases), whose gradient is the plain sum of the upstream gradient d over the broadcast axes,
    dG/db[c] = sum_{f,h} d[f,c,h], the array  S[c] = sum_{f,h} |d[f,c,h]|  (same shape as b); and for every LayerNorm SCALE g,
    dG/dg[c] = sum_{f,h} d[f,c,h] xhat[f,c,h], the array  S[c] = sum_{f,h} |d[f,c,h] xhat[f,c,h]|; and for every WEIGHT tensor W of a
    linear layer (conv / conv_transpose / dense kernels, the two merge matrices, the embedding table), dG/dW = sum_{f,h} x d, the same
    sum over |x| |d| (the layer's own weight-gradient operator applied to the absolute values).  All 44 trainables have an entry.  Such a sum cancels towards 0
    as a fit converges, so it has no scale of its own: the error of ANY finite-precision evaluation of it is bounded by
    (relative error of a term) x S[c], never by a fraction of |dG/db|.  S is the scale a comparison of these tensors is
    measured on where the gradient itself has cancelled away (tests/test_gpu_frame.py)."""
    torch = _torch()
    if kink is not None:
        kink = {k: (v if k == 'tau' else torch.as_tensor(np.asarray(v)).reshape(v.shape[0], v.shape[1], v.shape[2], 1))
                for k, v in kink.items()}
    dtype = dtype or torch.float32
    P = torch_params(P_np, dtype, requires_grad=True)
    xt = torch.tensor(np.asarray(x), dtype=dtype)
    yt = torch.tensor(np.asarray(y), dtype=torch.int64)
    et = torch.tensor(np.asarray(eps), dtype=dtype)
    tape = {} if sum_scales else None
    L = torch_loss(arch, P, xt, yt, et, kink, tape)
    L['G'].backward()
    grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v)).numpy().copy())
                        for k, v in P.items())
    out = {k: v.detach().numpy().copy() for k, v in L.items()}
    if not sum_scales:
        return out, grads
    scales = OrderedDict()
    lin = tape.pop('__lin__', [])
    for name, fn, xin, outt in lin:
        Wl = P[name].detach().clone().requires_grad_(True)
        (sw,) = torch.autograd.grad(fn(xin.abs(), Wl), Wl, grad_outputs=outt.grad.abs())
        scales[name] = sw.numpy().reshape(P[name].shape)
    for k, (t, factor) in tape.items():
        d = (t.grad if factor is None else t.grad * factor).abs()
        if d.dim() == 4:      # [F,C,H,1] -> per channel
            scales[k] = d.sum(dim=(0, 2, 3)).numpy().reshape(P[k].shape)
        else:                 # [F,N] -> per column
            scales[k] = d.sum(dim=0).numpy().reshape(P[k].shape)
    return out, grads, scales


def tf_adam_step(p, g, m, v, t, lr=1e-4, b1=0.5, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update (trainer/vae.py:16-24), t 1-based.
    Operates on arrays of any float dtype; returns (p, m, v)."""
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + eps)
    return p, m, v


# --------------------------------------------------------------------------
# data plane pieces (analyzer.py) restated in NumPy
# --------------------------------------------------------------------------
SP_DIM = 513
FEAT_DIM = SP_DIM * 2 + 3      # analyzer.py:19-22


def tanhize_forward(x, xmin, xmax):      # analyzer.py:77-84
    x = (x - xmin) / (xmax - xmin)
    return np.clip(x, 0.0, 1.0) * 2.0 - 1.0


def tanhize_backward(x, xmin, xmax):     # analyzer.py:86-87
    return (x * 0.5 + 0.5) * (xmax - xmin) + xmin


def parse_records(raw_bytes):            # analyzer.py:138-158
    v = np.frombuffer(raw_bytes, dtype='<f4').reshape(-1, FEAT_DIM)
    return dict(sp=v[:, :SP_DIM], ap=v[:, SP_DIM:2 * SP_DIM], f0=v[:, 2 * SP_DIM],
                en=v[:, 2 * SP_DIM + 1], speaker=v[:, 2 * SP_DIM + 2].astype(np.int64))


def convert_f0(f0, mu_s, std_s, mu_t, std_t):   # convert.py:51-57 (thresholds on the transformed value)
    f0 = np.asarray(f0, np.float32)
    lf0 = np.where(f0 > 1.0, np.log(np.where(f0 > 1.0, f0, 1.0)), f0).astype(np.float32)
    lf0 = np.where(lf0 > 1.0, (lf0 - np.float32(mu_s)) / np.float32(std_s) * np.float32(std_t) + np.float32(mu_t), lf0)
    lf0 = np.where(lf0 > 1.0, np.exp(lf0), lf0)
    return lf0.astype(np.float32)


def pw2wav_inputs(sp, ap, f0, en):       # analyzer.py:160-171 (dict branch, as convert.py:105-112 calls it)
    """Arrays handed to pyworld.synthesize: arithmetic in the input dtype, then float64 C-order copies."""
    en = np.reshape(en, [-1, 1])
    spl = en * np.power(10., sp)
    return (np.asarray(f0).astype(np.float64).copy(order='C'), spl.astype(np.float64).copy(order='C'),
            np.asarray(ap).astype(np.float64).copy(order='C'))
