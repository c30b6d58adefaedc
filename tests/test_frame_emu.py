"""The small-batch frame kernels (vae-npvc_amd/csrc/gfx950_frame.h) emulated on the host against the float64 oracle.

The kernel source is written against a phase runner, so tests/frame_emu/frame_emu.cpp compiles THE SAME per-thread code
with g++ (a phase = a loop over the 1024 thread ids).  What this pins on the CPU, before any GPU run: every tiling and
halo, the tap algebra of the strided / transposed convs in both directions, the rotating register window of the
1025-tap layer, the K-split reductions, the LayerNorm backward with its per-channel sums, and the packed weight copies.
(Not a product path: nothing under vae-npvc_amd/ loads this library.)
"""
import ctypes as C
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
    hdr = os.path.join(ROOT, 'vae-npvc_amd', 'csrc', 'gfx950_frame.h')
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
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


def run_emu(lib, arch, P, x, y, eps, mode=31, bwd=True, target=None, z_in=None, wgrad=False):
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
                           int(arch['y_dim']), F, mode, (2 if wgrad else 1) if bwd else 0, fptr(ws), fptr(toff))
    assert rc == 0

    def t(i, shape):
        return ws[toff[i]:toff[i] + int(np.prod(shape))].reshape(shape).astype(np.float64)
    return t, ws, toff


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_forward_pass_matches_the_oracle(emu, arch):
    F, seed = 3, 4
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    t, ws, toff = run_emu(emu, arch, P, x, y, eps, bwd=False)
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


def test_backward_pass_matches_autograd(emu, arch):
    F, seed = 3, 9
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    t, ws, toff = run_emu(emu, arch, P, x, y, eps)
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
    t, ws, toff = run_emu(emu, arch, P, x, y, eps, wgrad=True)
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
