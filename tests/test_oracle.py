"""Pins the CPU oracle: two independent implementations must agree, analytic
known-answer tests must hold, and the committed golden fixtures must be reproduced."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, SMALL_ARCH, load_arch, rel_err, sample_idx
from oracle import convvae_oracle as O


def test_shape_chain_and_pads(arch):
    g = O.geometry(arch)
    assert [l['hout'] for l in g['enc']] == [171, 57, 19, 7, 3]
    assert [l['pad'] for l in g['enc']] == [2, 2, 2, 3, 3]
    assert [l['hout'] for l in g['dec']] == [57, 171, 513, 513]
    assert [l['pad'] for l in g['dec']] == [3, 2, 2, 512]
    assert g['flat'] == 768 and g['merge'] == 1539
    L = O.param_layout(arch)
    assert len(L) == 44
    assert sum(int(np.prod(s)) for s in L.values()) == 939162


@pytest.mark.parametrize('which', ['vcc', 'small'])
def test_numpy_vs_torch_f64(which):
    arch = load_arch() if which == 'vcc' else SMALL_ARCH
    P = O.init_params(arch, 3)
    x, y, eps = O.make_inputs(arch, 3, 3)
    R = O.np_forward(arch, P, x, y, eps)
    Pt = O.torch_params(P, torch.float64)
    Lt = O.torch_loss(arch, Pt, torch.tensor(x, dtype=torch.float64), torch.tensor(y),
                      torch.tensor(eps, dtype=torch.float64))
    for k in ('z_mu', 'z_lv', 'xh'):
        assert rel_err(Lt[k].numpy(), R[k]) < 1e-10
    for k in ('G', 'D_KL', 'logP'):
        assert abs(float(Lt[k]) - R[k]) < 1e-10 * max(1.0, abs(R[k]))


def test_conv_transpose_is_adjoint_of_conv():
    # <conv(x), u> == <x, convT(u)> with the same kernel (TF: conv2d_transpose is the
    # input-gradient of conv2d); checks both SAME pad rules against each other.
    rng = np.random.default_rng(0)
    for (h, k, s, ci, co) in [(57, 7, 3, 3, 2), (19, 9, 3, 2, 3), (18, 4, 3, 2, 2), (513, 1025, 1, 1, 2)]:
        hin = h * s
        W = rng.standard_normal((k, 1, ci, co))         # conv layout [k,1,Cin,Cout]
        x = rng.standard_normal((2, ci, hin))
        u = rng.standard_normal((2, co, h))
        lhs = (O.np_conv_same(x, W, np.zeros(co), s) * u).sum()
        # the same array read as conv_transpose layout [k,1,Cout_T=ci,Cin_T=co]
        rhs = (x * O.np_convT_same(u, W, np.zeros(ci), s)).sum()
        assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))


def test_known_answers(arch):
    # GaussianKLD(0,0,0,0) = 128*0.5*(1/(1+1e-6) - 1)  (keeps the epsilon)
    P = O.init_params(arch, 0)
    z = np.zeros((1, 128))
    kld = 0.5 * ((0.0 - z) + (np.exp(z) + z ** 2) / (1 + 1e-6) - 1.0)
    assert abs(kld.sum() - 128 * 0.5 * (1 / (1 + 1e-6) - 1)) < 1e-12
    # GaussianLogDensity(x, x, 0) = -0.5*513*log(2 pi)
    assert abs(-0.5 * 513 * O.LOG_2PI + 471.4154) < 1e-3
    assert O.np_lrelu(np.array([-1.0]))[0] == -0.02
    # LN of a constant sample -> beta
    a = np.full((1, 4, 5), 3.25)
    out = O.np_layernorm(a, np.arange(4.0), np.ones(4))
    assert np.allclose(out[0, :, 0], np.arange(4.0), atol=1e-9)
    # all-zero weights and biases -> xh = 0 and logP closed form
    Z = {k: np.zeros_like(v) for k, v in P.items()}
    for k in Z:
        if k.endswith('.scale'):
            Z[k] = np.ones_like(Z[k])
    x, y, eps = O.make_inputs(arch, 2, 0)
    R = O.np_forward(arch, Z, x, y, eps)
    assert np.abs(R['xh']).max() == 0.0
    want = -0.5 * (513 * O.LOG_2PI + (x.astype(np.float64) ** 2).sum(1).mean() / (1 + 1e-6))
    assert abs(R['logP'] - want) < 1e-9
    # delta-kernel conv = strided copy
    W = np.zeros((7, 1, 1, 1)); W[2, 0, 0, 0] = 1.0     # pad_lo = 2 -> tap 2 is the centre
    xx = np.arange(513.0).reshape(1, 1, 513)
    out = O.np_conv_same(xx, W, np.zeros(1), 3)
    assert np.array_equal(out[0, 0], xx[0, 0, ::3])


def test_decode_depends_only_on_row_y(arch):
    P = O.init_params(arch, 1)
    z = np.random.default_rng(0).standard_normal((2, 128))
    y = np.array([3, 3])
    base = O.np_decode(arch, P, z, y)
    P2 = dict(P)
    e = P['y_embedding/y_emb'].copy()
    e[[0, 1, 2, 4, 5, 6, 7, 8, 9]] += 1.0
    P2['y_embedding/y_emb'] = e
    assert np.array_equal(O.np_decode(arch, P2, z, y), base)
    e = P['y_embedding/y_emb'].copy(); e[3] += 1.0
    P2['y_embedding/y_emb'] = e
    assert not np.array_equal(O.np_decode(arch, P2, z, y), base)


def test_autograd_matches_finite_differences():
    arch = SMALL_ARCH
    P = O.init_params(arch, 2)
    x, y, eps = O.make_inputs(arch, 3, 2)
    _, G = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64)
    rng = np.random.default_rng(0)
    for name in P:
        flat = P[name].astype(np.float64).ravel()
        for i in rng.choice(flat.size, size=min(2, flat.size), replace=False):
            h = 1e-6
            def f(delta):
                Q = {k: v.astype(np.float64) for k, v in P.items()}
                q = Q[name].ravel().copy(); q[i] += delta; Q[name] = q.reshape(P[name].shape)
                return O.np_forward(arch, Q, x, y, eps)['G']
            fd = (f(h) - f(-h)) / (2 * h)
            an = G[name].ravel()[i]
            assert abs(fd - an) < 1e-5 * max(1.0, abs(an)), (name, i, fd, an)


def test_tf_adam_first_step():
    g = np.array([0.5, -2.0, 1e-3])
    p, m, v = O.tf_adam_step(np.zeros(3), g, np.zeros(3), np.zeros(3), 1)
    # step 1: delta = -lr*sqrt(1-b2)/(1-b1) * g(1-b1)/(sqrt((1-b2)g^2)+1e-8) ~= -lr*sign(g)
    assert np.allclose(p, -1e-4 * np.sign(g), rtol=1e-3)
    lr_t = 1e-4 * math.sqrt(1 - 0.999) / (1 - 0.5)
    assert np.allclose(p, -lr_t * (0.5 * g) / (np.sqrt(0.001 * g * g) + 1e-8))


@pytest.mark.parametrize('fixture,which,F,seed', [('vcc2016_F4_seed0.npz', 'vcc', 4, 0),
                                                   ('small_F5_seed1.npz', 'small', 5, 1)])
def test_golden_fixtures_reproduced(fixture, which, F, seed):
    arch = load_arch() if which == 'vcc' else SMALL_ARCH
    gold = np.load(os.path.join(GOLDEN, fixture))
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    R = O.np_forward(arch, P, x, y, eps)
    for k in ('z_mu', 'z_lv', 'xh'):
        assert rel_err(R[k], gold[k]) < 1e-12
    assert np.allclose([R['G'], R['D_KL'], R['logP']], gold['loss3'], rtol=1e-12)
    # float32 torch path vs float64 golden: establishes the fp32 noise floor of the spec
    L32, G32 = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float32)
    assert rel_err(L32['xh'], gold['xh']) < 1e-4
    for i, n in enumerate(G32):
        assert abs(np.sqrt((G32[n].astype(np.float64) ** 2).sum()) - gold['grad_l2'][i]) < 2e-4 * max(gold['grad_l2'][i], 1e-6)


def test_data_plane_restatement():
    rng = np.random.default_rng(0)
    xmin = rng.uniform(-12, -8, 513).astype(np.float32)
    xmax = xmin + rng.uniform(2, 6, 513).astype(np.float32)
    sp = rng.uniform(-14, -2, (6, 513)).astype(np.float32)
    x = O.tanhize_forward(sp, xmin, xmax)
    assert x.min() >= -1 and x.max() <= 1 and (x == -1).any() and (x == 1).any()
    inside = (sp > xmin) & (sp < xmax)
    back = O.tanhize_backward(x, xmin, xmax)
    assert np.allclose(back[inside], sp[inside], atol=1e-4)
    rec = np.zeros((3, O.FEAT_DIM), np.float32)
    rec[:, -1] = [0, 9, 4]
    rec[:, :513] = sp[:3]
    d = O.parse_records(rec.tobytes())
    assert d['speaker'].dtype == np.int64 and d['speaker'].tolist() == [0, 9, 4]
    assert np.array_equal(d['sp'], sp[:3])
    f0 = np.array([0.0, 100.0, 250.0, 0.5], np.float32)
    out = O.convert_f0(f0, 5.0, 0.2, 4.8, 0.3)
    assert out[0] == 0.0 and out[3] == 0.5
    assert np.allclose(out[1], np.exp((np.log(100.0) - 5.0) / 0.2 * 0.3 + 4.8), rtol=1e-5)


def test_philox_block_function_known_answers():
    """Philox4x32-10 against the Random123 known-answer vectors (kat_vectors: counter, key -> output); the device
    sampler (csrc/philox.h) is compared with this restatement in tests/test_gpu_runtime.py."""
    from oracle import philox_ref as P
    kats = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
            ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
            ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
             [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for c, k, want in kats:
        got = P.philox4x32_10(np.array([c], np.uint32), np.array([k], np.uint32))[0]
        assert [int(v) for v in got] == want
    x = P.normal(1 << 18, seed=11, offset=3)
    assert x.dtype == np.float32 and abs(x.mean()) < 1e-2 and abs(x.std() - 1) < 1e-2 and abs((x ** 4).mean() - 3) < 0.1
    assert not np.array_equal(x[:64], P.normal(64, seed=11, offset=4)) and np.array_equal(x[:64], P.normal(64, 11, 3))


def test_lrelu_kink_branch_pin():
    """oracle.torch_lrelu with pinned branches: pinning every unit to its own side changes nothing; pinning an
    ambiguous unit (|n| < tau) to the other side changes the gradient by that unit's slope difference -- the finite
    jump a float32 evaluation can legitimately make when its rounding error crosses the kink."""
    import torch
    from helpers import SMALL_ARCH
    arch = SMALL_ARCH
    P = O.init_params(arch, 4)
    x, y, eps = O.make_inputs(arch, 6, 4)
    _, G0 = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64)
    g = O.geometry(arch)
    Pt = O.torch_params(P, torch.float64)
    _, _, acts = O.torch_encode(arch, Pt, torch.tensor(x, dtype=torch.float64))
    kink = {'tau': 1e-4}
    ns = {}
    for i, (a, cur) in enumerate(acts):
        p = 'Encoder/Conv2d-%d/' % i
        n = O.torch_layernorm(a, Pt[p + 'layernorm.offset'], Pt[p + 'layernorm.scale']).numpy()[..., 0]
        ns['enc%d' % i] = n
        kink['enc%d' % i] = n >= 0
    z_mu, z_lv, _ = O.torch_encode(arch, Pt, torch.tensor(x, dtype=torch.float64))
    z = z_mu + torch.tensor(eps, dtype=torch.float64) * torch.sqrt(torch.exp(z_lv))
    _, dacts = O.torch_decode(arch, Pt, z, torch.tensor(y))
    for i, a in enumerate(dacts[1:]):
        n = O.torch_layernorm(a, Pt['Generator/ConvT-LN%d.offset' % i], Pt['Generator/ConvT-LN%d.scale' % i]).numpy()[..., 0]
        kink['dec%d' % i] = n >= 0
    _, G1 = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64, kink=kink)
    for n in G0:
        assert np.array_equal(G0[n], G1[n])
    # flip the branch of the unit closest to the kink, with tau just above its |n|
    n0 = ns['enc0']
    idx = np.unravel_index(np.abs(n0).argmin(), n0.shape)
    k2 = dict(kink)
    k2['tau'] = float(np.abs(n0[idx])) * 1.5
    k2['enc0'] = kink['enc0'].copy()
    k2['enc0'][idx] = not k2['enc0'][idx]
    _, G2 = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64, kink=k2)
    d = max(np.abs(G2[n] - G0[n]).max() / np.abs(G0[n]).max() for n in G0)
    assert d > 1e-6, d          # one flipped unit of ~300 moves the gradient visibly


def test_sum_of_magnitudes_scales_bound_every_gradient():
    """oracle.torch_loss_and_grads(sum_scales=True): for each of the 44 trainables the scale S = sum over (frame, position) of |term| of the
    sum its gradient is.  Pinned here: every tensor has one, of its own shape; S >= |gradient| entry by entry (triangle inequality: a replay
    function that is not the layer's weight-gradient operator breaks it); the gradients themselves are unchanged by the tape; and for ONE
    frame with the upstream gradient of one sign the additive parameters of the last layer have S == |gradient|."""
    import torch
    arch = load_arch()
    P = O.init_params(arch, 3)
    x, y, eps = O.make_inputs(arch, 16, 3)
    _, G, S = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64, sum_scales=True)
    _, G0 = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64)
    assert list(S.keys()).__len__() == 44 and set(S) == set(G)
    for k in G:
        assert np.array_equal(G[k], G0[k])
        assert S[k].shape == G[k].shape and np.all(S[k] >= np.abs(G[k]) * (1 - 1e-9)), k
    b = 'Generator/conv2d_transpose_3/bias'
    assert S[b].ravel()[0] > abs(G[b].ravel()[0])            # 16 frames x 513 residuals of both signs: cancellation
    x1 = np.full((1, 513), 100.0)                            # one frame far above anything the untrained decoder emits: every residual has one sign
    _, G1, S1 = O.torch_loss_and_grads(arch, P, x1, y[:1], eps[:1], torch.float64, sum_scales=True)
    assert abs(S1[b].ravel()[0] - abs(G1[b].ravel()[0])) <= 1e-12 * S1[b].ravel()[0]
