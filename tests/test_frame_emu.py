"""The small-batch frame kernels (vae-npvc_amd/csrc/gfx950_frame.h) emulated on the host against the float64 oracle.

The kernel source is written against a phase runner, so tests/frame_emu/frame_emu.cpp compiles THE SAME per-thread code
with g++ (a phase = a loop over the 1024 thread ids).  What this pins on the CPU, before any GPU run: every tiling and
halo, the tap algebra of the strided / transposed convs in both directions, the rotating register window of the
1025-tap layer, the K-split reductions, the LayerNorm backward with its per-channel sums, and the packed weight copies.
(Not a product path: nothing under vae-npvc_amd/ loads this library.)
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from helpers import ROOT, load_arch
from oracle import convvae_oracle as O

EMU_DIR = os.path.join(ROOT, 'tests', 'frame_emu')
TOL = 2e-5          # fp32 FMA chains against float64, relative to the tensor's largest entry


@pytest.fixture(scope='module')
def emu():
    so = os.path.join(EMU_DIR, 'libframe_emu.so')
    src = os.path.join(EMU_DIR, 'frame_emu.cpp')
    hdrs = [os.path.join(ROOT, 'vae-npvc_amd', 'csrc', h) for h in ('gfx950_frame.h', 'gfx950_frame_wgrad.h', 'disc_frame.h')]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-o', so, src])
    lib = C.CDLL(so)
    lib.frame_emu_run.restype = C.c_int
    return lib


def poff_table(arch):
    """POff of gfx950_frame.h as an int array (order of make_off in frame_emu.cpp) from the oracle's layout"""
    lay, offs, off = O.param_layout(arch), {}, 0
    for n, shp in lay.items():
        offs[n] = off
        off += int(np.prod(shp))
    t = [offs['y_embedding/y_emb']]
    for i in range(5):
        p = 'Encoder/Conv2d-%d/' % i
        t += [offs[p + 'kernel'], offs[p + 'bias'], offs[p + 'layernorm.offset'], offs[p + 'layernorm.scale']]
    t += [offs['Encoder/dense/kernel'], offs['Encoder/dense/bias'], offs['Encoder/dense_1/kernel'], offs['Encoder/dense_1/bias']]
    t += [offs['Generator/fully_connected/weights'], offs['Generator/fully_connected/biases'],
          offs['Generator/fully_connected_1/weights'], offs['Generator/fully_connected_1/biases'], offs['Generator/BiasAdd/biases']]
    for i in range(4):
        p = 'Generator/conv2d_transpose%s/' % ('' if i == 0 else '_%d' % i)
        t += [offs[p + 'kernel'], offs[p + 'bias']]
        if i < 3:
            t += [offs['Generator/ConvT-LN%d.offset' % i], offs['Generator/ConvT-LN%d.scale' % i]]
    assert len(t) == 44
    return np.array(t, np.int32), offs, off


ENC_N = [2736, 1824, 1216, 896, 768]
DEC_N = [1824, 2736, 4104]


def workspace(lib, F):
    sizes = ([F * n for n in ENC_N] + [2 * F] * 5 + [F * 128] * 4 + [F * 1539] + [F * n for n in DEC_N] + [2 * F] * 3
             + [F * 513, F, F, F * 513] + [F * n for n in DEC_N] + [F * 1539, F * 128, F * 128] + [F * n for n in ENC_N]
             + [F * 3 * lib.frame_emu_lnp_c(), lib.frame_emu_pack_floats(), 939162, F * (12000 + 4104)])
    assert len(sizes) == lib.frame_emu_tensor_count()
    offs = np.concatenate([[0], np.cumsum([(s + 63) // 64 * 64 for s in sizes])]).astype(np.int64)
    ws = np.full(int(offs[-1]), np.nan, np.float32)
    g0 = offs[len(sizes) - 2]
    ws[g0:g0 + sizes[-2]] = 0.0            # the gradient buffer is zero-filled by the pack launch
    return ws, offs[:-1].copy(), sizes


def fptr(a):
    return a.ctypes.data_as(C.c_void_p)


def run_emu(lib, arch, P, x, y, eps, mode=31, bwd=True, target=None, z_in=None, wgrad=False, split=False):
    F = x.shape[0]
    po, _, _ = poff_table(arch)
    flat = O.flatten_params(P)
    ws, toff, sizes = workspace(lib, F)
    T_PK = lib.frame_emu_tensor_count() - 3
    pk = ws[toff[T_PK]:toff[T_PK] + sizes[T_PK]]
    lib.frame_emu_pack(fptr(flat), fptr(po), fptr(pk))
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.int64)
    eps = None if eps is None else np.ascontiguousarray(eps, np.float32)
    rc = lib.frame_emu_run(fptr(flat), fptr(po), fptr(x), fptr(target) if target is not None else None, fptr(y),
                           fptr(eps) if eps is not None else None, fptr(z_in) if z_in is not None else None,
                           int(arch['y_dim']), F, mode, ((2 if wgrad else 1) if bwd else 0) + (4 if split else 0), fptr(ws), fptr(toff))
    assert rc == 0

    def t(i, shape):
        return ws[toff[i]:toff[i] + int(np.prod(shape))].reshape(shape).astype(np.float64)
    return t, ws, toff


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize('split', [False, True])
def test_forward_pass_matches_the_oracle(emu, arch, split):
    """split: the 1025-tap layer and the log-density from the eight-workgroups-per-frame bodies (toep_split_fwd), as in a
    train step; otherwise inside the frame kernel (encode / decode / loss entry points)"""
    F, seed = 3, 4
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    t, ws, toff = run_emu(emu, arch, P, x, y, eps, bwd=False, split=split)
    R = O.np_forward(arch, P, x, y, eps)
    g = O.geometry(arch)
    errs = {}
    for i, l in enumerate(g['enc']):
        a = R['enc_a%d' % i]
        errs['enc_a%d' % i] = rel(t(i, a.shape), a)
        st = t(5 + i, (F, 2))
        errs['enc_mean%d' % i] = float(np.abs(st[:, 0] - a.mean(axis=(1, 2))).max() / np.abs(a).max())
        errs['enc_rstd%d' % i] = rel(st[:, 1], 1 / np.sqrt(a.var(axis=(1, 2)) + 1e-5))
    for k, i in (('z_mu', 10), ('z_lv', 11), ('z', 12), ('h', 14)):
        errs[k] = rel(t(i, R[k].shape), R[k])
    assert np.array_equal(t(13, (F, 128)), eps.astype(np.float64))          # the draw is handed to the backward pass
    for i in range(3):
        a = R['dec_a%d' % i]
        errs['dec_a%d' % i] = rel(t(15 + i, a.shape), a)
        errs['dec_rstd%d' % i] = rel(t(18 + i, (F, 2))[:, 1], 1 / np.sqrt(a.var(axis=(1, 2)) + 1e-5))
    errs['xh'] = rel(t(21, R['xh'].shape), R['xh'])
    kl, nll = t(22, (F,)), t(23, (F,))
    errs['D_KL'] = abs(kl.mean() - R['D_KL']) / abs(R['D_KL'])
    errs['logP'] = abs(nll.mean() - R['logP']) / abs(R['logP'])
    if not split:      # (the split step leaves d(xh) to its backward half)
        dxh = (R['xh'] - x.astype(np.float64)) / (1 + 1e-6) / F
        errs['d_xh'] = rel(t(24, dxh.shape), dxh)
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad


def test_decode_only_and_encode_only_modes(emu, arch):
    """conversion path (convert.py:79-89): encode -> z_mu, decode(z_mu, target speaker)"""
    F, seed = 2, 6
    P = O.init_params(arch, seed)
    x, y, _ = O.make_inputs(arch, F, seed)
    t, _, _ = run_emu(emu, arch, P, x, y, None, mode=1, bwd=False)
    R = O.np_forward(arch, P, x, y, None)
    assert rel(t(10, (F, 128)), R['z_mu']) < TOL
    z = np.ascontiguousarray(R['z_mu'], np.float32)
    t2, _, _ = run_emu(emu, arch, P, x, y, None, mode=4, bwd=False, z_in=z)
    assert rel(t2(21, (F, 513)), R['xh']) < TOL


def oracle_intermediate_grads(arch, P_np, x, y, eps):
    """float64 autograd with the gradients of the INTERMEDIATE tensors retained (pre-LN conv outputs, h, z_mu, z_lv, xh)"""
    P = O.torch_params(P_np, torch.float64, requires_grad=True)
    xt, yt, et = torch.tensor(x, dtype=torch.float64), torch.tensor(y), torch.tensor(eps, dtype=torch.float64)
    z_mu, z_lv, eacts = O.torch_encode(arch, P, xt)
    z = z_mu + et * torch.sqrt(torch.exp(z_lv))
    xh, dacts = O.torch_decode(arch, P, z, yt)
    keep = {'z_mu': z_mu, 'z_lv': z_lv, 'xh': xh, 'h': dacts[0]}
    for i, (a, _) in enumerate(eacts):
        keep['enc_a%d' % i] = a
    for i, a in enumerate(dacts[1:]):
        keep['dec_a%d' % i] = a
    for v in keep.values():
        v.retain_grad()
    kld = 0.5 * ((0.0 - z_lv) + (torch.exp(z_lv) + z_mu ** 2) / (1.0 + O.EPSILON) - 1.0)
    lp = -0.5 * (O.LOG_2PI + (xt.reshape(xt.shape[0], -1) - xh) ** 2 / (1.0 + O.EPSILON))
    G = -lp.sum(-1).mean() + kld.sum(-1).mean()
    G.backward()
    return {k: v.grad.numpy().reshape(v.shape[0], -1) for k, v in keep.items()}, {k: v.grad.numpy() for k, v in P.items()}


@pytest.mark.parametrize('split', [False, True])
def test_backward_pass_matches_autograd(emu, arch, split):
    F, seed = 3, 9
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    t, ws, toff = run_emu(emu, arch, P, x, y, eps, split=split)
    G, GP = oracle_intermediate_grads(arch, P, x, y, eps)
    errs = {}
    errs['d_xh'] = rel(t(24, (F, 513)), G['xh'])
    for i, n in enumerate(DEC_N):
        errs['d_dec_a%d' % i] = rel(t(25 + i, (F, n)), G['dec_a%d' % i])
    errs['d_h'] = rel(t(28, (F, 1539)), G['h'])
    errs['d_z_mu'] = rel(t(29, (F, 128)), G['z_mu'])
    errs['d_z_lv'] = rel(t(30, (F, 128)), G['z_lv'])
    for i, n in enumerate(ENC_N):
        errs['d_enc_a%d' % i] = rel(t(31 + i, (F, n)), G['enc_a%d' % i])
    # per-channel sums of the LayerNorm backward, reduced over frames here = gradients of offset / scale / conv bias
    lnp = t(36, (F, 3, emu.frame_emu_lnp_c())).sum(axis=0)
    layers = [('Generator/ConvT-LN2', 'Generator/conv2d_transpose_2/bias', 8), ('Generator/ConvT-LN1', 'Generator/conv2d_transpose_1/bias', 16),
              ('Generator/ConvT-LN0', 'Generator/conv2d_transpose/bias', 32)]
    layers += [('Encoder/Conv2d-%d/layernorm' % i, 'Encoder/Conv2d-%d/bias' % i, c) for i, c in ((4, 256), (3, 128), (2, 64), (1, 32), (0, 16))]
    off = 0
    for ln, bias, c in layers:
        errs[ln + '.offset'] = rel(lnp[0, off:off + c], GP[ln + '.offset'].ravel())
        errs[ln + '.scale'] = rel(lnp[1, off:off + c], GP[ln + '.scale'].ravel())
        errs[bias] = rel(lnp[2, off:off + c], GP[bias].ravel())
        off += c
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad


@pytest.mark.parametrize('F,seed', [(3, 12), (37, 13), (66, 14)])
def test_one_launch_weight_gradient_matches_autograd(emu, arch, F, seed):
    """the job list of the weight-gradient launch, block by block, after the emulated passes: all 44 gradients
    (F = 37: more frames than the smaller jobs have frame chunks, ragged chunks; F = 66: the two-slice form of the 1025-tap job)"""
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    t, ws, toff = run_emu(emu, arch, P, x, y, eps, wgrad=True, split=(F != 37))
    _, GP = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64)
    T_G = emu.frame_emu_tensor_count() - 2
    g = ws[toff[T_G]:toff[T_G] + 939162].astype(np.float64)
    assert np.isfinite(g).all()
    _, offs, n = poff_table(arch)
    assert n == g.size
    bad = {}
    for name, want in GP.items():
        got = g[offs[name]:offs[name] + want.size].reshape(want.shape)
        e = rel(got, want)
        if not e < 5e-5:
            bad[name] = e
    assert not bad, bad


# ---------------------------------------------------------------------------------------------- critic front path
def test_critic_front_kernels_against_float64_autograd(emu):
    """csrc/disc_frame.h (the VAWGAN critic's two thin conv layers: per-row pass kernels + the job-list weight gradient)
    emulated on the host and chained as vaenpvc_disc_critic_fwd_bwd chains them, with the 115-tap layer cut out: its input
    gradients are random inputs (c2 for pass 2 on the rows xi, c4 for pass 4 on all rows).  With a1 = block(rows) the two
    conv + LayerNorm + lrelu layers, the chain computes the gradient of
        L = sum_rows <c4, a1>  +  lambda * mean_f (|g_f| - 1)^2 ,   g = d(sum_{xi rows} <c2, a1>) / d(rows xi)
    with respect to the layers' eight parameter tensors, and  at1 = dL/dc2  (what the chain hands back up to the 115-tap
    layer in pass 3).  Float64 double-backward autograd of the same expression is the reference."""
    import torch
    from oracle import vawgan_oracle as V
    arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vawgan-vcc2016.json')))
    F, B, lam = 3, 9, 10.0
    rng = np.random.RandomState(5)
    D = V.disc_init_params(arch, 3)
    names = ['Discriminator/Conv2d-%d/%s' % (i, k) for i in range(2) for k in ('kernel', 'bias', 'layernorm.scale', 'layernorm.offset')]
    flat, off = [], {}
    for n in names:
        off[n] = sum(len(v) for v in flat)
        flat.append(np.asarray(D[n], np.float64).reshape(-1))
    P = np.concatenate(flat).astype(np.float32)
    rows = np.tanh(rng.randn(B, 513)).astype(np.float32)
    c2 = (0.05 * rng.randn(F, 32 * 57)).astype(np.float32)
    c4 = (0.05 * rng.randn(B, 32 * 57)).astype(np.float32)
    # ---- float64 reference
    Dt = {n: torch.tensor(np.asarray(D[n], np.float64), requires_grad=True) for n in names}
    g0 = V.disc_geometry(arch)[:2]

    def block(x):
        cur = x.reshape(x.shape[0], 1, -1, 1)
        outs = []
        for i, l in enumerate(g0):
            p = 'Discriminator/Conv2d-%d/' % i
            a = O.torch_conv_same(cur, Dt[p + 'kernel'], Dt[p + 'bias'], l['s'])
            outs.append(a)
            cur = O.torch_lrelu(O.torch_layernorm(a, Dt[p + 'layernorm.offset'], Dt[p + 'layernorm.scale']))
        return cur.reshape(x.shape[0], -1), outs
    xt = torch.tensor(rows.astype(np.float64))
    xi = xt[2 * F:].clone().requires_grad_(True)
    c2t = torch.tensor(c2.astype(np.float64), requires_grad=True)
    c4t = torch.tensor(c4.astype(np.float64))
    a1_all, pre = block(xt)
    a1_xi, _ = block(xi)
    g, = torch.autograd.grad((c2t * a1_xi).sum(), xi, create_graph=True)
    nrm = torch.sqrt((g ** 2).sum(-1))
    L = (c4t * a1_all).sum() + lam * ((nrm - 1.0) ** 2).mean()
    ref = torch.autograd.grad(L, [Dt[n] for n in names] + [c2t])
    # ---- emulated kernels
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    off8 = np.array([off[names[0]], off[names[1]], off[names[2]], off[names[3]], off[names[4]], off[names[5]], off[names[6]], off[names[7]]],
                    np.int32)
    u0, st0 = np.zeros((B, 16 * 171), np.float32), np.zeros((B, 2), np.float32)
    u1, st1 = np.zeros((B, 32 * 57), np.float32), np.zeros((B, 2), np.float32)
    ain2, gg, gpf = np.zeros((B, 32 * 57), np.float32), np.zeros((F, 513), np.float32), np.zeros(F, np.float32)
    at1 = np.zeros((F, 32 * 57), np.float32)
    grads = np.zeros(112 + 3584 + 16 + 32 + 16 + 16 + 32 + 32, np.float32)
    emu.critic_front_emu.restype = C.c_int
    rc = emu.critic_front_emu(fp(P), off8.ctypes.data_as(C.POINTER(C.c_int)), fp(rows), F, fp(c2), fp(c4), C.c_float(2.0 * lam / F),
                              fp(u0), fp(st0), fp(u1), fp(st1), fp(ain2), fp(gg), fp(gpf), fp(at1), fp(grads))
    assert rc == 0
    errs = {}
    errs['u0'] = rel(u0, pre[0].detach().numpy().reshape(B, -1))
    errs['u1'] = rel(u1, pre[1].detach().numpy().reshape(B, -1))
    errs['a1'] = rel(ain2, a1_all.detach().numpy())
    errs['g'] = rel(gg, g.detach().numpy())
    errs['gp_f'] = rel(gpf, ((nrm - 1.0) ** 2).detach().numpy())
    errs['at1'] = rel(at1, ref[8].numpy())
    pos = 0
    # order of the grads buffer: dW0 dW1 db0 db1 dgamma0 dbeta0 dgamma1 dbeta1
    for key, n in (('kernel0', 112), ('kernel1', 3584), ('bias0', 16), ('bias1', 32), ('scale0', 16), ('offset0', 16), ('scale1', 32),
                   ('offset1', 32)):
        idx = {'kernel0': 0, 'bias0': 1, 'scale0': 2, 'offset0': 3, 'kernel1': 4, 'bias1': 5, 'scale1': 6, 'offset1': 7}[key]
        errs['d ' + key] = rel(grads[pos:pos + n], ref[idx].numpy().reshape(-1))
        pos += n
    bad = {k: v for k, v in errs.items() if not v < 5e-5}
    assert not bad, bad
