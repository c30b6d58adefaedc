"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle.

Bars: integer plumbing bit-exact; activations / losses / converted frames
<= 1e-4 relative (||a-b||_inf / max(||b||_inf, 1e-6)) against the float64 oracle;
gradients and Adam trajectories <= 2e-4 (fp32 summation over the batch).
Every measured error is appended to gpurun_out/parity_report.txt.
"""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, ROOT, SMALL_ARCH, golden_large_inputs, load_arch, rel_err, sample_idx
from oracle import convvae_oracle as O

pytestmark = pytest.mark.gpu

TOL_ACT = 1e-4
TOL_GRAD = 2e-4       # ONE gradient bar for every batch size
REPORT = os.path.join(ROOT, 'gpurun_out', 'parity_report.txt')


def report(tag, err, tol):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, 'a') as fp:
        fp.write('%-70s err=%.3e tol=%.1e %s\n' % (tag, err, tol, 'OK' if err <= tol else 'FAIL'))


def check(tag, got, want, tol, fails):
    e = rel_err(got, want)
    report(tag, e, tol)
    if not (e <= tol):
        fails.append('%s: %.3e > %.1e' % (tag, e, tol))


ARCHS = {'vcc': load_arch(), 'small': SMALL_ARCH}


FRAME_BIT = 1 << 21      # whole-frame-per-workgroup kernels for batches <= 512 (gfx950_frame.h); cleared = layered kernels


def make_engine(which, impl, masks=(0xffffffff, 0xffffffff), precision=None, frame=False):
    """frame=False (default here): the LAYERED kernels at every batch size -- the tests of this file pin each layered
    kernel family at small batch sizes; the small-batch frame kernels have their own tests (tests/test_gpu_frame.py)."""
    from hipvae import Engine
    eng = Engine(ARCHS[which], impl=impl, precision=precision)
    f, b = (masks[0], masks[1]) if frame else (masks[0] & ~FRAME_BIT, masks[1] & ~FRAME_BIT)
    eng.set_tuned_masks(f, b)      # state of THIS engine's context (nothing process-global)
    return eng


def upload(eng, P, x, y, eps):
    eng.load_flat(O.flatten_params(P))
    dev = eng.device
    return (torch.tensor(x, device=dev), torch.tensor(y, device=dev), torch.tensor(eps, device=dev))


KINK_TAU = 1e-4


def gpu_branches(eng, arch, P, F):
    """Which side of the lrelu kink the GPU forward pass took, per LayerNorm output: recomputed in float64 from the
    GPU's own pre-LN tensors and statistics in the workspace.  The gradient oracle pins the units with |n| < KINK_TAU
    to these branches (oracle.torch_lrelu): the comparison is then about arithmetic, not about which side of a kink a
    rounding error fell on (measured without the pin: a 2-term split evaluation at F = 37 flips ~4 of 685k units and
    every flip moves a gradient tensor by ~1/F of its scale)."""
    from hipvae import lib as L
    g = O.geometry(arch)
    out = {'tau': KINK_TAU}
    for net, layers, pre in (('enc', g['enc'], 'Encoder/Conv2d-%d/layernorm'), ('dec', g['dec'][:-1], 'Generator/ConvT-LN%d')):
        for i, l in enumerate(layers):
            a = eng.ws_region(F, L.MODE_TRAIN, '%s_a%d' % (net, i)).cpu().numpy().astype(np.float64).reshape(F, l['cout'], l['hout'])
            st = eng.ws_region(F, L.MODE_TRAIN, '%s_st%d' % (net, i)).cpu().numpy().astype(np.float64).reshape(F, 2)
            gam = np.asarray(P[(pre % i) + '.scale'], np.float64).reshape(1, -1, 1)
            bet = np.asarray(P[(pre % i) + '.offset'], np.float64).reshape(1, -1, 1)
            n = (a - st[:, 0].reshape(F, 1, 1)) * st[:, 1].reshape(F, 1, 1) * gam + bet
            out['%s%d' % (net, i)] = n >= 0
    return out


def oracle_grads(eng, arch, P, x, y, eps):
    """float64 autograd oracle with the kink branches of the run that just finished on `eng`"""
    return O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64, kink=gpu_branches(eng, arch, P, x.shape[0]))


def run_train(eng, P, x, y, eps):
    xt, yt, et = upload(eng, P, x, y, eps)
    grads = torch.full((eng.n_params,), float('nan'), device=eng.device)
    l3 = eng.train_fwd_bwd(xt, yt, et, grads).clone()
    torch.cuda.synchronize()
    return l3.cpu().numpy(), grads.cpu().numpy()


CASES = [('small', 'generic', 5, 1), ('vcc', 'generic', 4, 0), ('vcc', 'auto', 4, 0), ('vcc', 'auto', 37, 5),
         ('small', 'auto', 5, 1)]


@pytest.mark.parametrize('which,impl,F,seed', CASES)
def test_forward_intermediates_and_losses(which, impl, F, seed):
    from hipvae import lib as L
    arch = ARCHS[which]
    eng = make_engine(which, impl)
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    l3, _ = run_train(eng, P, x, y, eps)
    R = O.np_forward(arch, P, x, y, eps)
    g = O.geometry(arch)
    fails = []
    tag = '%s/%s/F%d ' % (which, impl, F)

    def region(name):
        return eng.ws_region(F, L.MODE_TRAIN, name).cpu().numpy()
    for i, l in enumerate(g['enc']):
        a = R['enc_a%d' % i]
        check(tag + 'enc_a%d' % i, region('enc_a%d' % i).reshape(a.shape), a, TOL_ACT, fails)
        st = region('enc_st%d' % i).reshape(F, 2)
        mu = a.mean(axis=(1, 2)); rstd = 1 / np.sqrt(a.var(axis=(1, 2)) + 1e-5)
        # the mean is compared on the scale of the activations (it may legitimately be ~0)
        check(tag + 'enc_mean%d' % i, st[:, 0] / np.abs(a).max(), mu / np.abs(a).max(), TOL_ACT, fails)
        check(tag + 'enc_rstd%d' % i, st[:, 1], rstd, TOL_ACT, fails)
    for k in ('z_mu', 'z_lv', 'z', 'h'):
        check(tag + k, region(k).reshape(R[k].shape), R[k], TOL_ACT, fails)
    for i in range(len(g['dec']) - 1):
        a = R['dec_a%d' % i]
        check(tag + 'dec_a%d' % i, region('dec_a%d' % i).reshape(a.shape), a, TOL_ACT, fails)
        st = region('dec_st%d' % i).reshape(F, 2)
        check(tag + 'dec_rstd%d' % i, st[:, 1], 1 / np.sqrt(a.var(axis=(1, 2)) + 1e-5), TOL_ACT, fails)
    check(tag + 'xh', region('xh').reshape(R['xh'].shape), R['xh'], TOL_ACT, fails)
    check(tag + 'loss3', l3, np.array([R['G'], R['D_KL'], R['logP']]), TOL_ACT, fails)
    report(tag + 'recon-L1 mean|xh-xh_ref|', float(np.abs(region('xh').reshape(R['xh'].shape) - R['xh']).mean()), 1.0)
    assert not fails, '\n'.join(fails)


@pytest.mark.parametrize('which,impl,F,seed', CASES + [('vcc', 'auto', 256, 2)])
def test_gradients(which, impl, F, seed):
    arch = ARCHS[which]
    eng = make_engine(which, impl)
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    l3, grads = run_train(eng, P, x, y, eps)
    assert np.isfinite(grads).all(), 'some gradient entries were never written'
    L, G = oracle_grads(eng, arch, P, x, y, eps)
    fails = []
    tag = '%s/%s/F%d grad ' % (which, impl, F)
    for name, (off, shape) in eng.layout.items():
        n = int(np.prod(shape))
        check(tag + name, grads[off:off + n].reshape(shape), G[name], TOL_GRAD, fails)
    check(tag + 'loss3', l3, np.array([L['G'], L['D_KL'], L['logP']]), TOL_ACT, fails)
    assert not fails, '\n'.join(fails)


@pytest.mark.parametrize('which,impl,fixture,F,seed', [('vcc', 'generic', 'vcc2016_F4_seed0.npz', 4, 0),
                                                        ('vcc', 'auto', 'vcc2016_F4_seed0.npz', 4, 0),
                                                        ('small', 'generic', 'small_F5_seed1.npz', 5, 1)])
def test_against_committed_golden_vectors(which, impl, fixture, F, seed):
    arch = ARCHS[which]
    gold = np.load(os.path.join(GOLDEN, fixture))
    eng = make_engine(which, impl)
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    fails = []
    tag = 'golden %s/%s ' % (which, impl)
    xt, yt, et = upload(eng, P, x, y, eps)
    from hipvae.dp import Stepper
    st = Stepper(eng, 1e-4, 0.5, 0.999)
    idx = sample_idx(eng.n_params, 64)
    p0 = eng.params.cpu().numpy().astype(np.float64)[idx]
    for t in (1, 2, 3):
        l3 = st.step(xt, yt, et).clone()
        if t == 1:
            from hipvae import lib as L
            check(tag + 'loss3', l3.cpu().numpy(), gold['loss3'], TOL_ACT, fails)
            for k in ('z_mu', 'z_lv', 'xh'):
                got = eng.ws_region(F, L.MODE_TRAIN, k).cpu().numpy().reshape(gold[k].shape)
                check(tag + k, got, gold[k], TOL_ACT, fails)
            g = st.grads.cpu().numpy()
            for i, (name, (off, shape)) in enumerate(eng.layout.items()):
                n = int(np.prod(shape))
                l2 = np.sqrt((g[off:off + n].astype(np.float64) ** 2).sum())
                e = abs(l2 - gold['grad_l2'][i]) / max(gold['grad_l2'][i], 1e-6)
                report(tag + 'grad_l2 ' + name, e, TOL_GRAD)
                if e > TOL_GRAD:
                    fails.append('grad_l2 %s %.3e' % (name, e))
                k = min(8, n)
                e = np.abs(g[off:off + n][sample_idx(n)] - gold['grad_samples'][i][:k]).max() / max(gold['grad_absmax'][i], 1e-6)
                report(tag + 'grad_samples ' + name, e, TOL_GRAD)
                if e > TOL_GRAD:
                    fails.append('grad_samples %s %.3e' % (name, e))
        # Adam trajectory: compare the UPDATE (p_t - p_0), which is what the optimiser computes
        got = eng.params.cpu().numpy().astype(np.float64)[idx] - p0
        want = gold['adam_p%d' % t] - p0
        e = np.abs(got - want).max() / np.abs(want).max()
        report(tag + 'adam step %d (delta)' % t, e, 2e-3)
        if e > 2e-3:
            fails.append('adam step %d: %.3e' % (t, e))
    assert not fails, '\n'.join(fails)


@pytest.mark.parametrize('which,impl,F', [('vcc', 'generic', 9), ('vcc', 'auto', 9), ('vcc', 'auto', 1),
                                          ('vcc', 'auto', 700), ('small', 'generic', 3)])
def test_encode_decode_conversion_path(which, impl, F):
    """convert.py:79-89: x -> encode (z_mu) -> decode(target id)."""
    arch = ARCHS[which]
    eng = make_engine(which, impl)
    P = O.init_params(arch, 7)
    x, y, _ = O.make_inputs(arch, F, 7)
    eng.load_flat(O.flatten_params(P))
    xt = torch.tensor(x, device=eng.device)
    z_mu, z_lv = eng.encode(xt.view(F, 1, -1, 1), want_lv=True)
    trg = arch['y_dim'] - 1
    yt = torch.full((F,), trg, dtype=torch.int64, device=eng.device)
    xh = eng.decode(z_mu, yt)
    R = O.np_forward(arch, P, x, np.full(F, trg), None)
    fails = []
    tag = 'convert %s/%s/F%d ' % (which, impl, F)
    check(tag + 'z_mu', z_mu.cpu().numpy(), R['z_mu'], TOL_ACT, fails)
    check(tag + 'z_lv', z_lv.cpu().numpy(), R['z_lv'], TOL_ACT, fails)
    check(tag + 'xh', xh.cpu().numpy(), R['xh'], TOL_ACT, fails)
    assert not fails, '\n'.join(fails)


def test_data_plane_kernels(arch):
    """Tanhize fwd/bwd (analyzer.py:82-87) and record unpacking with the bit-exact
    float32 -> int64 speaker cast (analyzer.py:127)."""
    eng = make_engine('vcc', 'auto')
    rng = np.random.default_rng(0)
    xmin = rng.uniform(-12, -8, 513).astype(np.float32)
    xmax = xmin + rng.uniform(2, 6, 513).astype(np.float32)
    N = 1000
    rec = rng.standard_normal((N, 1029)).astype(np.float32)
    rec[:, :513] = rng.uniform(-14, -2, (N, 513))
    spk = rng.integers(0, 10, N)
    rec[:, -1] = spk.astype(np.float32)
    dev = eng.device
    tmin, tmax = torch.tensor(xmin, device=dev), torch.tensor(xmax, device=dev)
    x, y = eng.unpack_records(torch.tensor(rec, device=dev), tmin, tmax)
    assert y.dtype == torch.int64 and np.array_equal(y.cpu().numpy(), spk.astype(np.int64))
    want = O.tanhize_forward(rec[:, :513].astype(np.float64), xmin.astype(np.float64), xmax.astype(np.float64))
    assert np.abs(x.cpu().numpy() - want).max() < 2e-6
    assert x.min().item() >= -1.0 and x.max().item() <= 1.0
    back = eng.tanhize(x, tmin, tmax, forward=False).cpu().numpy()
    wantb = O.tanhize_backward(want, xmin.astype(np.float64), xmax.astype(np.float64))
    assert rel_err(back, wantb) < 1e-6
    fw = eng.tanhize(torch.tensor(rec[:, :513], device=dev), tmin, tmax, forward=True)
    assert torch.equal(fw, x)


@pytest.mark.parametrize('impl', ['generic', 'auto'])
def test_properties_at_full_batch(arch, impl):
    """Size-independent properties at the metric's batch (F = 256)."""
    eng = make_engine('vcc', impl)
    F = 256
    P = O.init_params(arch, 11)
    x, y, eps = O.make_inputs(arch, F, 11)
    l3, g = run_train(eng, P, x, y, eps)
    # (1) data-parallel identity: grads of the full batch == mean of the grads of the two halves
    h = F // 2
    la, ga = run_train(eng, P, x[:h], y[:h], eps[:h])
    lb, gb = run_train(eng, P, x[h:], y[h:], eps[h:])
    assert rel_err(0.5 * (ga + gb), g) < 1e-4
    assert np.allclose(0.5 * (la + lb), l3, rtol=1e-5)
    # (2) frame independence: permuting frames permutes outputs
    perm = np.random.default_rng(0).permutation(F)
    xt = torch.tensor(x, device=eng.device)
    z = eng.encode(xt).cpu().numpy()
    zp = eng.encode(torch.tensor(x[perm], device=eng.device)).cpu().numpy()
    assert rel_err(zp, z[perm]) < 1e-6
    # (3) decode(z, y) depends on the embedding only through row y
    zt = torch.tensor(z, device=eng.device)
    y3 = torch.full((F,), 3, dtype=torch.int64, device=eng.device)
    base = eng.decode(zt, y3).clone()
    views = eng.param_views()
    views['y_embedding/y_emb'][[0, 1, 2, 4, 5, 6, 7, 8, 9]] += 1.0
    assert torch.equal(eng.decode(zt, y3), base)
    views['y_embedding/y_emb'][3] += 1.0
    assert not torch.equal(eng.decode(zt, y3), base)
    # (4) all-zero weights -> xh = 0 and closed-form logP
    eng.params.zero_()
    for name, v in eng.param_views().items():
        if name.endswith('.scale'):
            v.fill_(1.0)
    et = torch.tensor(eps, device=eng.device)
    yt = torch.tensor(y, device=eng.device)
    l3z = eng.loss_fwd(xt, yt, et).cpu().numpy()
    want = -0.5 * (513 * O.LOG_2PI + (x.astype(np.float64) ** 2).sum(1).mean() / (1 + 1e-6))
    assert abs(l3z[2] - want) < 1e-4 * abs(want)
    # float32 evaluation of 1/(1+1e-6) - 1 (what TF float32 would also produce): -9.5367e-07 per dim
    want_kl = 128 * 0.5 * float(np.float32(1.0) / (np.float32(1.0) + np.float32(1e-6)) - np.float32(1.0))
    assert abs(l3z[1] - want_kl) < 1e-6 and abs(l3z[1] - 128 * 0.5 * (1 / (1 + 1e-6) - 1)) < 5e-6


def test_argument_errors(arch):
    from hipvae import HipVaeError
    eng = make_engine('vcc', 'auto')
    x = torch.zeros(4, 513, device=eng.device)
    with pytest.raises(TypeError):
        eng.encode(x.double())
    with pytest.raises(ValueError):
        eng.encode(torch.zeros(4, 512, device=eng.device))
    with pytest.raises(TypeError):
        eng.decode(torch.zeros(4, 128, device=eng.device), torch.zeros(4, dtype=torch.int32, device=eng.device))
    # workspace too small -> VAENPVC_E_WORKSPACE through the ABI
    import ctypes as C
    from hipvae import lib as L
    z = torch.zeros(4, 128, device=eng.device)
    ws = torch.zeros(1024, dtype=torch.uint8, device=eng.device)
    rc = eng.lib.vaenpvc_encode_fwd(eng.ctx, eng.params.data_ptr(), x.data_ptr(), 4, z.data_ptr(), None,
                                    ws.data_ptr(), 1024, None)
    assert rc == -2 and b'workspace too small' in eng.lib.vaenpvc_last_error()


# ---------------------------------------------------------------------------------------
# Tuned gfx950 kernels, one step at a time: only the named step runs the tuned kernel, every
# other step runs the geometry-generic kernel, so a failure names the faulty kernel.
STEPS = ['e0', 'e1', 'e2', 'e3', 'e4', 'heads', 'merge', 'd0', 'd1', 'd2', 'd3']
_oracle_cache = {}


def oracle_case(F, seed):
    key = (F, seed)
    if key not in _oracle_cache:
        arch = ARCHS['vcc']
        P = O.init_params(arch, seed)
        x, y, eps = O.make_inputs(arch, F, seed)
        R = O.np_forward(arch, P, x, y, eps)
        _oracle_cache[key] = (P, x, y, eps, R)
    return _oracle_cache[key]


def compare_everything(eng, F, seed, tag, tol_grad=TOL_GRAD):
    from hipvae import lib as L
    arch = ARCHS['vcc']
    P, x, y, eps, R = oracle_case(F, seed)
    l3, grads = run_train(eng, P, x, y, eps)
    _, G = oracle_grads(eng, arch, P, x, y, eps)
    fails = []
    g = O.geometry(arch)

    def region(name, like):
        return eng.ws_region(F, L.MODE_TRAIN, name).cpu().numpy().reshape(like.shape)
    for i in range(len(g['enc'])):
        check(tag + 'enc_a%d' % i, region('enc_a%d' % i, R['enc_a%d' % i]), R['enc_a%d' % i], TOL_ACT, fails)
        a = R['enc_a%d' % i]
        st = eng.ws_region(F, L.MODE_TRAIN, 'enc_st%d' % i).cpu().numpy().reshape(F, 2)
        check(tag + 'enc_rstd%d' % i, st[:, 1], 1 / np.sqrt(a.var(axis=(1, 2)) + 1e-5), TOL_ACT, fails)
    for k in ('z_mu', 'z_lv', 'z', 'h'):
        check(tag + k, region(k, R[k]), R[k], TOL_ACT, fails)
    for i in range(len(g['dec']) - 1):
        check(tag + 'dec_a%d' % i, region('dec_a%d' % i, R['dec_a%d' % i]), R['dec_a%d' % i], TOL_ACT, fails)
        a = R['dec_a%d' % i]
        st = eng.ws_region(F, L.MODE_TRAIN, 'dec_st%d' % i).cpu().numpy().reshape(F, 2)
        check(tag + 'dec_rstd%d' % i, st[:, 1], 1 / np.sqrt(a.var(axis=(1, 2)) + 1e-5), TOL_ACT, fails)
    check(tag + 'xh', region('xh', R['xh']), R['xh'], TOL_ACT, fails)
    check(tag + 'loss3', l3, np.array([R['G'], R['D_KL'], R['logP']]), TOL_ACT, fails)
    if not np.isfinite(grads).all():
        fails.append(tag + 'non-finite / unwritten gradient entries')
    for name, (off, shape) in eng.layout.items():
        n = int(np.prod(shape))
        check(tag + 'grad ' + name, grads[off:off + n].reshape(shape), G[name], tol_grad, fails)
    return fails


@pytest.mark.parametrize('direction', ['fwd', 'bwd'])
@pytest.mark.parametrize('step', STEPS)
def test_tuned_step_isolated(step, direction):
    bit = 1 << STEPS.index(step)
    masks = (bit, 0) if direction == 'fwd' else (0, bit)
    eng = make_engine('vcc', 'auto', masks)
    fails = compare_everything(eng, 37, 5, 'isolated %s/%s ' % (direction, step))
    assert not fails, '\n'.join(fails)


@pytest.mark.parametrize('F,seed', [(37, 5), (64, 6), (1, 7), (33, 8)])
def test_all_tuned_steps(F, seed):
    eng = make_engine('vcc', 'auto')
    fails = compare_everything(eng, F, seed, 'tuned F%d ' % F)
    assert not fails, '\n'.join(fails)


BF16_TOEP = (0xbfffffff, 0xffffffff)   # forward-mask bit 30 cleared: bf16-split Toeplitz kernels at any batch size


@pytest.mark.parametrize('precision', ['bf16x2', 'bf16x3'])
@pytest.mark.parametrize('F,seed', [(37, 5), (64, 6), (1, 7), (130, 9)])
def test_bf16_split_toeplitz_kernels_against_oracle(F, seed, precision):
    """The last decoder layer on the bf16 matrix cores (operands split into 2 or 3 bf16 terms, fp32
    accumulation): forward, input gradient and weight gradient against the float64 oracle, same
    tolerances as the fp32 kernels.  (By default these kernels only run at F >= 8192; the mask forces them.)"""
    eng = make_engine('vcc', 'auto', BF16_TOEP, precision=precision)
    fails = compare_everything(eng, F, seed, '%s toeplitz F%d ' % (precision, F))
    assert not fails, '\n'.join(fails)


PLANE_GEMM = (0xefffffff, 0xefffffff)   # bit 28 of both masks cleared: plane GEMM kernels at any batch size


@pytest.mark.parametrize('precision', ['bf16x2', 'bf16x3'])
@pytest.mark.parametrize('F,seed', [(37, 5), (130, 9), (1, 7), (257, 12)])
def test_plane_gemm_dense_layers_against_oracle(F, seed, precision):
    """Encoder heads, merge FC and encoder layer 4 (as a dense layer) on the bf16 matrix cores
    (csrc/gfx950_planegemm.h: operand planes, C = A B^T forward / input gradient, C += A^T B weight gradient with
    the 7-tap fold for layer 4) against the float64 oracle at batch sizes with ragged 128-row tiles.
    (By default these kernels run at F >= 1024; the mask forces them.)"""
    eng = make_engine('vcc', 'auto', PLANE_GEMM, precision=precision)
    fails = compare_everything(eng, F, seed, '%s plane-gemm F%d ' % (precision, F))
    assert not fails, '\n'.join(fails)


VIEW_CONV_BITS = 0xebffffff            # (= VIEW_CONV below: plane GEMMs + every conv site on the view GEMMs at any batch size)


@pytest.mark.parametrize('F,seed', [(1, 7), (37, 5), (130, 9), (257, 12), (1027, 7)])
def test_ring_gemm_dense_layers_against_oracle(F, seed, monkeypatch):
    """Round 6: C = A B^T of the K-long dense-shaped sites (encoder layer 4 as a dense layer, forward + input gradient; the heads, forward +
    input gradient) on the four-wave LDS-DMA ring kernel (csrc/gfx950_ntring.h: 256 x 128 tiles, 128 x 64 wave tiles, per-(operand, plane)
    64-k slots, three product phases per chunk with all four k-steps of a plane's fragments in registers), forced with VAENPVC_NT_RING=2 at
    ragged batches (one 256-row tile holding 1 / 37 / 130 rows, two tiles with 1 row in the second, five with 3): every tensor and gradient
    against the float64 oracle; the kernel must have run (the six C = A B^T sites are then split between it and k_gemm_nt)."""
    monkeypatch.setenv('VAENPVC_NT_RING', '2')
    monkeypatch.setenv('VAENPVC_CG_SF_RING', '2')
    monkeypatch.setenv('VAENPVC_CG_PF_RING', '2')        # ... and its input gradient + layer 2's LayerNorm backward on the (3, 3) instance (24 frames per tile)        # encoder layer 3 forward on the same main loop (36 whole frames per tile)
    eng = make_engine('vcc', 'auto', masks=(PLANE_GEMM[0] & VIEW_CONV_BITS, PLANE_GEMM[1] & VIEW_CONV_BITS), precision='bf16x2')
    fails = compare_everything(eng, F, seed, 'nt_ring F%d ' % F)
    assert not fails, '\n'.join(fails)


BF16_TOL_ACT, BF16_TOL_GRAD = 3e-2, 6e-2   # bf16 MODE (one bf16 term per operand, ~3 significant digits)
@pytest.mark.parametrize('F,seed', [(128, 3), (384, 4), (1152, 5)])
def test_a_resident_merge_gemm_against_oracle(F, seed, monkeypatch):
    """The merge layer's forward GEMM on the A-resident kernel (k_gemm_nt_ar, round 5: one workgroup owns 128 frames, their K run stays
    in LDS, the 13 column tiles of Wz stream past it; the default from 24 576 frames on) forced at small batches (VAENPVC_NT_AR=2,
    plane GEMMs at any batch size): every tensor and gradient against the float64 oracle -- h, everything behind it, and the ragged
    last column tile (1 539 = 12 x 128 + 3)."""
    monkeypatch.setenv('VAENPVC_NT_AR', '2')
    eng = make_engine('vcc', 'auto', masks=PLANE_GEMM)
    fails = compare_everything(eng, F, seed, 'nt_ar F%d ' % F)
    assert not fails, '\n'.join(fails)


RAGGED_LARGE_TENSOR_BAR = 1e-3     # two GPU evaluations of an UNFILTERED batch differ by their lrelu kink flips: measured worst
                                   # tensor 4.9e-4 (encoder layer 0's 112-entry kernel, downstream of every flip), y_emb 2.7e-4, the large weight
                                   # tensors <= 1.6e-4; a wrong or missing frame in a ragged tile moves a tensor by far more


def test_tuned_and_generic_paths_agree_at_a_ragged_large_batch():
    """20 011 frames (not a multiple of 16, 18, 64 or 128): the frame-owning GEMM tiles (k_cgemm_pf: 16 frames, k_cgemm_sf: 18), the plane GEMMs'
    128-row tiles, the fused layer kernels' groups and the 1025-tap kernels' 64-frame tiles all end in a ragged tile at a size the float64
    fixtures (8 192 / 32 768) do not cover and the in-test oracle cannot reach.  The generic kernels (one thread per output, any geometry; pinned
    against float64 at small sizes) are the reference: losses and all 44 gradient tensors of the tuned default selection against them."""
    from hipvae import Engine
    arch = ARCHS['vcc']
    F, seed = 20011, 31
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    res = {}
    for impl in ('generic', 'auto'):
        eng = Engine(arch, impl=impl)
        res[impl] = run_train(eng, P, x, y, eps)
        layout = eng.layout
        del eng
    (l_g, g_g), (l_t, g_t) = res['generic'], res['auto']
    assert np.isfinite(g_t).all() and np.isfinite(g_g).all()
    fails, tag = [], 'tuned vs generic F%d ' % F
    check(tag + 'loss3', l_t, l_g, 1e-5, fails)
    for name, (off, shape) in layout.items():
        n = int(np.prod(shape))
        check(tag + 'grad ' + name, g_t[off:off + n], g_g[off:off + n], RAGGED_LARGE_TENSOR_BAR, fails)
    assert not fails, '\n'.join(fails)


def test_decoder_tail_in_the_forward_epilogue(monkeypatch):
    """k_fconv<TAIL> (round 5; built, measured, OFF by default because it is not faster): decoder layer 2's forward kernel with the work of
    the pass behind it in its epilogue -- LayerNorm statistics of its result, the 1025-tap layer's operand planes, bin 512 of the activated
    tensor, output column 512.  Forced with VAENPVC_D2_TAIL=1 at a ragged large batch (odd number of frames: the last group holds one frame):
    every tensor and gradient against the float64 oracle."""
    monkeypatch.setenv('VAENPVC_D2_TAIL', '1')
    eng = make_engine('vcc', 'auto')
    eng.timer_select('dec2_stats_planes')
    fails = compare_everything(eng, 1027, 7, 'd2_tail F1027 ')
    _, n = eng.timer_read()
    eng.timer_select(None)
    assert n == 0, 'the separate statistics / planes pass still ran'
    assert not fails, '\n'.join(fails)


@pytest.mark.parametrize('precision', ['bf16x2'])
@pytest.mark.parametrize('F,seed', [(37, 7), (257, 12), (1027, 7)])
def test_layernorm_on_load_in_the_tap_layer_forward(F, seed, precision, monkeypatch):
    """Round 6: no pass between decoder layer 2 and the 1025-tap layer.  Decoder layer 2's forward kernel leaves the LayerNorm statistics of its
    result (k_fconv<..., OST>: per-wave (count, sum, centred squares) combined by Chan's formula), the 1025-tap forward kernel stages the fp32
    tensor itself -- LayerNorm + lrelu + operand split on the way into LDS -- and also emits the weight-gradient kernel's operand planes, output
    column 512 and bin 512 of the activated tensor (k_toep_gemm_bf16<..., LNA>).  Default from 16 384 frames per step on (one channel group per
    frame tile); forced here with VAENPVC_D2_LNA=2 at ragged batches (last frame tile with 37 / 1 / 3 frames, last 2-frame group with one frame):
    every tensor and gradient against the float64 oracle, and the separate pass must not have run.  (Two operand planes and fewer: the kernel
    keeps two A tiles in LDS, three planes do not fit; the bf16 mode runs it by default in the 32 768-frame fixture test.)"""
    monkeypatch.setenv('VAENPVC_D2_LNA', '2')
    eng = make_engine('vcc', 'auto', FUSED_CONV if F < 1024 else (0xffffffff, 0xffffffff), precision=precision)
    eng.timer_select('dec2_stats_planes')
    fails = compare_everything(eng, F, seed, 'd2_lna %s F%d ' % (precision, F))
    _, n = eng.timer_read()
    eng.timer_select(None)
    assert n == 0, 'the separate statistics / planes pass still ran'
    assert not fails, '\n'.join(fails)


def test_the_separate_plane_producer_pass_still_serves(monkeypatch):
    """VAENPVC_D2_LNA=0: the round-5 form (k_ln_stats_act_planes between decoder layer 2 and the 1025-tap layer) stays the A/B partner and the
    path of the batch sizes with several channel groups per frame tile: same comparison at a ragged large batch."""
    monkeypatch.setenv('VAENPVC_D2_LNA', '0')
    eng = make_engine('vcc', 'auto')
    eng.timer_select('dec2_stats_planes')
    fails = compare_everything(eng, 1027, 7, 'd2_lna=0 F1027 ')
    _, n = eng.timer_read()
    eng.timer_select(None)
    assert n == 1, 'the separate statistics / planes pass did not run'
    assert not fails, '\n'.join(fails)


@pytest.mark.parametrize('skip', ['1', '0'])
def test_loss_kernel_with_and_without_the_fp32_copy_of_its_gradient(monkeypatch, skip):
    """Round 6: from 1 024 frames on the loss kernel (k_nll_dxh_post) hands d(xh) to both GEMMs of the 1025-tap layer as bf16 planes; its fp32
    copy in the workspace tensor `d_xh` has no reader then and is no longer stored (VAENPVC_DXH_SKIP=1, default).  Both settings at a ragged
    large batch: every tensor and gradient against the float64 oracle; with 0 the tensor holds -(x - xh) / ((1 + 1e-6) F) (model/vae.py:120-125,
    util/layers.py:159-167), with 1 it is left untouched."""
    from hipvae import lib as L
    monkeypatch.setenv('VAENPVC_DXH_SKIP', skip)
    eng = make_engine('vcc', 'auto')
    F, seed = 1027, 7
    eng.ws_region(F, L.MODE_TRAIN, 'd_xh').fill_(123.0)
    fails = compare_everything(eng, F, seed, 'dxh_skip=%s F%d ' % (skip, F))
    P, x, y, eps, R = oracle_case(F, seed)
    d = eng.ws_region(F, L.MODE_TRAIN, 'd_xh').cpu().numpy().reshape(F, 513)
    if skip == '0':
        want = -(np.asarray(x, np.float64).reshape(F, 513) - R['xh'].reshape(F, 513)) / ((1.0 + 1e-6) * F)
        check('dxh_skip=0 d_xh', d, want, TOL_ACT, fails)
    elif not (d == 123.0).all():
        fails.append('dxh_skip=1: the fp32 copy of d(xh) was written')
    assert not fails, '\n'.join(fails)


@pytest.mark.parametrize('planes_out', ['1', '0'])
def test_merge_gradient_operand_from_the_layer_above(monkeypatch, planes_out):
    """Decoder layer 0's input-gradient kernel writes d(h) as the bf16 operand planes of the two merge GEMMs itself (k_fconv_r<..., POUT>,
    round 5, default from 1 024 frames on) and the per-speaker column sums are taken from those planes (k_segsum_planes); with
    VAENPVC_D0G_PLANES=0 it stores fp32 d(h) and k_split_segsum makes planes + sums in a pass of its own.  Both at a ragged large batch (the
    last group of the kernel holds three frames), every tensor and gradient against the float64 oracle -- the three merge biases, dWy and dE are
    functions of the column sums alone."""
    monkeypatch.setenv('VAENPVC_D0G_PLANES', planes_out)
    eng = make_engine('vcc', 'auto')
    fails = compare_everything(eng, 1027, 11, 'd(h) planes=%s F1027 ' % planes_out)
    assert not fails, '\n'.join(fails)


VIEW_CONV = (0xebffffff, 0xebffffff)    # ... and bit 26: every conv site on the view GEMMs (csrc/gfx950_viewconv.h)


@pytest.mark.parametrize('precision', ['bf16x2', 'bf16x3'])
@pytest.mark.parametrize('F,seed', [(37, 5), (130, 9), (1, 7), (257, 12)])
def test_view_conv_layers_against_oracle(F, seed, precision):
    """Encoder layers 1-3 and decoder layers 0-2 (forward, input gradient, weight gradient) as GEMMs over the
    overlapping-row view of channel-last bf16 planes (csrc/gfx950_viewconv.h) against the float64 oracle -- the
    kernels of the bf16 training mode, run here with 2 / 3 operand planes so the fp32-class bars apply and pin the
    indexing of every site (strided S-type, phase-stacked P-type, transposed weight-gradient epilogue)."""
    eng = make_engine('vcc', 'auto', VIEW_CONV, precision=precision)
    fails = compare_everything(eng, F, seed, '%s view-conv F%d ' % (precision, F))
    assert not fails, '\n'.join(fails)


FUSED_CONV = (0xfd3fffff, 0xfc3fffff)   # bits 25, 23, 22 (and 24, backward): the thin and medium conv sites / weight gradients on the fused kernels and
                                        # encoder layer 0 on its wave-per-frame kernels at any batch size


@pytest.mark.parametrize('precision', ['bf16x2', 'bf16x3'])
@pytest.mark.parametrize('F,seed', [(37, 5), (130, 9), (1, 7), (257, 12)])
def test_fused_thin_conv_layers_against_oracle(F, seed, precision):
    """Encoder layers 1-2 and decoder layers 1-2, forward / input gradient on the fused view-GEMM kernel
    (csrc/gfx950_fconv.h: fp32 frames converted to bf16 terms on their way into LDS, weights resident, no planes in
    HBM) and weight gradients on its counterpart (csrc/gfx950_fwgrad.h: both operands staged in LDS, the whole gradient
    tile in one workgroup's accumulators, one flush) against the float64 oracle; batch sizes with ragged 4-frame /
    2-frame workgroups.  (Three operand planes: the weight gradients and the medium site fall back.)"""
    eng = make_engine('vcc', 'auto', FUSED_CONV, precision=precision)
    fails = compare_everything(eng, F, seed, '%s fused-conv F%d ' % (precision, F))
    assert not fails, '\n'.join(fails)


FUSED_BWD = (0xffffffff, 0xffffbfff)    # bit 14 of the backward mask cleared: the one-kernel backward step of the thin decoder layers at any batch size


@pytest.mark.parametrize('layers', ['7', '1', '2', '4'])
@pytest.mark.parametrize('F,seed', [(37, 5), (130, 9), (1, 7), (257, 12), (600, 13)])
def test_fused_layer_backward_against_oracle(F, seed, layers, monkeypatch):
    """Decoder layers 2 and 1 and encoder layer 1: LayerNorm + lrelu backward, input gradient, weight gradient and the layer's
    d gamma / d beta / d bias in ONE kernel per layer (csrc/gfx950_fbwd.h: the gradient at the pre-LN output exists only as bf16
    terms in LDS; bits of VAENPVC_FB_LAYERS = decoder 2, decoder 1, encoder 1) against the float64 oracle: all three, and each
    alone next to the three-kernel form of its neighbours (the hand-over buffers differ; encoder layer 0's LayerNorm backward
    then runs in place); batch sizes below and above one frame per workgroup slot (512).  (By default from 1024 frames on.)"""
    monkeypatch.setenv('VAENPVC_FB_LAYERS', layers)
    eng = make_engine('vcc', 'auto', FUSED_BWD)
    eng.timer_select({'7': 'dec2_bwd,dec1_bwd,enc1_bwd', '1': 'dec2_bwd', '2': 'dec1_bwd', '4': 'enc1_bwd'}[layers])
    fails = compare_everything(eng, F, seed, 'fused-bwd layers=%s F%d ' % (layers, F))
    _, n = eng.timer_read()
    eng.timer_select(None)
    assert n == (3 if layers == '7' else 1), 'the fused backward kernels did not run'
    assert not fails, '\n'.join(fails)


@pytest.mark.parametrize('F,seed', [(37, 5), (257, 12), (600, 13)])
def test_fused_layer_backward_three_planes(F, seed, monkeypatch):
    """Round 6: the one-kernel backward step of the three thin layers with THREE operand planes (precision bf16x3, the fp32-exact mode
    that bench.py reports as reference_precision_value; until round 5 that mode fell back to the three-kernel form per layer, 2.2 ms
    of its step).  Same kernels, same bars; the plain 1e-4 / 2e-4 comparison against float64."""
    monkeypatch.setenv('VAENPVC_FB_LAYERS', '7')
    eng = make_engine('vcc', 'auto', FUSED_BWD, precision='bf16x3')
    eng.timer_select('dec2_bwd,dec1_bwd,enc1_bwd')
    fails = compare_everything(eng, F, seed, 'fused-bwd bf16x3 F%d ' % F)
    _, n = eng.timer_read()
    eng.timer_select(None)
    assert n == 3, 'the fused backward kernels did not run'
    assert not fails, '\n'.join(fails)


@pytest.mark.parametrize('precision', ['bf16x2', 'bf16x3', 'bf16'])
def test_default_selection_at_a_ragged_large_batch(precision):
    """The DEFAULT kernel selection just above its thresholds, at a batch size that is a multiple of nothing the kernels
    tile by (1027 frames: last frame groups of 3, 1 and 7 frames for the fused kernels' groups of 4, 2 and 8; 128-row
    GEMM tiles with 3 rows; 32-row reduction chunks with odd tails) against the float64 oracle run inside the test:
    every activation and every gradient, lrelu kink units pinned."""
    F, seed = 1027, 31
    eng = make_engine('vcc', 'auto', precision=precision)
    if precision == 'bf16':
        fails = []
        P, x, y, eps, R = oracle_case(F, seed)
        l3, grads = run_train(eng, P, x, y, eps)
        _, G = oracle_grads(eng, ARCHS['vcc'], P, x, y, eps)
        assert np.isfinite(grads).all()
        check('bf16 ragged F1027 loss3', l3, np.array([R['G'], R['D_KL'], R['logP']]), BF16_TOL_ACT, fails)
        for name, (off, shape) in eng.layout.items():
            n = int(np.prod(shape))
            check('bf16 ragged F1027 grad ' + name, grads[off:off + n].reshape(shape), G[name], BF16_TOL_GRAD, fails)
    else:
        fails = compare_everything(eng, F, seed, '%s ragged F%d ' % (precision, F))
    assert not fails, '\n'.join(fails)


def _golden_large(F, seed, precision, tag, tol_act=TOL_ACT, tol_grad=TOL_GRAD):
    """Default (auto) path at a benchmarked batch size against the committed chunked-float64 oracle fixture
    (tests/golden/make_golden_large.py): losses, z_mu / z_lv / xh rows of 16 sampled frames, and per tensor the
    gradient L2 norm and 64 sampled entries -- ONE gradient bar for every batch size."""
    from hipvae import lib as L
    arch = ARCHS['vcc']
    gold = np.load(os.path.join(GOLDEN, 'vcc2016_F%d_seed%d.npz' % (F, seed)))
    eng = make_engine('vcc', 'auto', precision=precision)
    P = O.init_params(arch, seed)
    x, y, eps = golden_large_inputs(arch, gold, F, seed)      # kink-safe frames of the seeded candidate stream
    l3, g = run_train(eng, P, x, y, eps)
    fails = []
    check(tag + 'loss3', l3, gold['loss3'], tol_act, fails)
    fidx = gold['frame_idx']
    for k, width in (('z_mu', 128), ('z_lv', 128), ('xh', 513)):
        got = eng.ws_region(F, L.MODE_TRAIN, k).view(F, width)[torch.as_tensor(fidx, device=eng.device)].cpu().numpy()
        check(tag + k + ' rows', got, gold[k + '_rows'], tol_act, fails)
    report(tag + 'recon-L1 mean|xh-xh_ref| (16 frames)',
           float(np.abs(eng.ws_region(F, L.MODE_TRAIN, 'xh').view(F, 513)[torch.as_tensor(fidx, device=eng.device)].cpu().numpy()
                        - gold['xh_rows']).mean()), 1.0)
    assert np.isfinite(g).all()
    for i, (name, (off, shape)) in enumerate(eng.layout.items()):
        n = int(np.prod(shape))
        gi = g[off:off + n].astype(np.float64)
        e = abs(np.sqrt((gi ** 2).sum()) - gold['grad_l2'][i]) / max(gold['grad_l2'][i], 1e-12)
        report(tag + 'grad_l2 ' + name, e, tol_grad)
        if e > tol_grad:
            fails.append('grad_l2 %s %.3e' % (name, e))
        k = min(64, n)
        e = np.abs(gi[sample_idx(n, 64)] - gold['grad_samples'][i][:k]).max() / max(gold['grad_absmax'][i], 1e-12)
        # ONE bar for every batch size (the fixture's frames are kink-safe: the float32 CPU restatement's own error on
        # these entries, recorded as ref32_grad_err, is <= 4e-7, so nothing but arithmetic is being compared)
        report(tag + 'grad_samples ' + name + ' (fp32 stand-in: %.1e)' % gold['ref32_grad_err'][i], e, tol_grad)
        if e > tol_grad:
            fails.append('grad_samples %s %.3e > %.3e' % (name, e, tol_grad))
    return fails


@pytest.mark.parametrize('F,seed', [(8192, 21), (32768, 22)])
@pytest.mark.parametrize('precision', ['bf16x2', 'bf16x3'])
def test_benchmarked_batch_sizes_against_oracle_fixture(F, seed, precision):
    """F = 32768 is the bench's batch (256 x [1,513,128]); F = 8192 the smallest one that selects the bf16-split
    Toeplitz kernels by default.  Both fp32-class precisions are held to the SAME bars as the small batches:
    1e-4 on activations / losses, 2e-4 on gradients (weight gradient of the 1025-tap layer with its frame-chunk
    split and atomics included)."""
    fails = _golden_large(F, seed, precision, 'golden F%d %s ' % (F, precision))
    assert not fails, '\n'.join(fails)




UNFILTERED_SHARE_SLACK = 2e-4     # share of the ~53 k sampled entries a path may have above the bar beyond the fp32 CPU restatement's share
# bounds of the DEFAULT precision (2-term operands) on unfiltered data: how often a LayerNorm output lands on the other side
# of the lrelu kink than in float64 grows with the error of the evaluation that produced it (1e-5 of the activation scale
# with 16-mantissa-bit operands against 1e-7 in float32), so this mode sees ~5x the kink-induced gradient error of a float32
# evaluation of the SAME batch: measured worst tensor 3.1e-4 (the speaker embedding; fp32 CPU restatement 5.5e-5), 9 of
# 52 850 sampled entries above 2e-4, median 3.2e-6.  Stated, not hidden: the fp32-exact mode (bf16x3) is held to the
# float32 restatement's own numbers.
UNFILTERED_X2_TENSOR_BAR = 3e-4   # round 5: was 5e-4; measured 2.2e-4 (y_emb; which term carries it: DESIGN.md section 5, measured by
                                  # test_kink_flips_explain_the_unfiltered_gradient_excess -- the flips, not the operand rounding)
UNFILTERED_X2_SHARE_BAR = 2.5e-4  # round 6: was 5e-4; measured 1.3e-4


@pytest.mark.parametrize('precision', ['bf16x3', 'bf16x2'])
def test_unfiltered_benchmark_batch_statistics(precision):
    """The benchmarked batch size on PLAIN seeded inputs (32 768 frames, lrelu kink units included; fixture
    tests/golden/make_golden_unfiltered.py).  A unit whose LayerNorm output lies within rounding of 0 takes slope 1 in
    one evaluation and 0.02 in another, so on such a batch no float32-class implementation can be held to a fixed
    max-norm gradient bar against float64.  The bound is therefore statistical and relative to
    what float32 arithmetic itself does on the SAME batch -- the fixture holds, for up to 4096 sampled entries of
    every gradient tensor, the float64 oracle and the oracle's float32 PyTorch-CPU restatement:
      (1) losses and the sampled z_mu / z_lv / xh rows: the ordinary 1e-4 bar (a kink does not move activations);
      (2) per tensor: max error on the sampled entries (relative to the tensor's largest entry)
          <= max(2e-4, the float32 CPU restatement's max error on the same entries)       [bf16x3, fp32-exact operands]
          <= UNFILTERED_X2_TENSOR_BAR                                                    [bf16x2, the default];
      (3) over all sampled entries: the share of entries off by more than 2e-4 of their tensor's scale is at most
          the float32 CPU restatement's share + UNFILTERED_SHARE_SLACK [bf16x3] / UNFILTERED_X2_SHARE_BAR [bf16x2];
      (4) the median error stays at rounding level (<= 2e-5): kinks are rare events, not a shift.
      (5) the kink flips themselves are COUNTED at this batch size: the fixture of make_golden_kink_units.py lists the
          ~50 k of the batch's 607 M lrelu units whose float64 LayerNorm output lies within 1e-4 of the kink (only those can
          flip: asserted at 2 048 frames by the mechanism test); the GPU's branch at exactly those units is recomputed in
          float64 from the GPU's own pre-LN values and statistics, and flips / all units <= KINK_FLIP_RATE_BAR.
    (The mechanism itself -- flips counted over ALL units, pinned, plain bar -- is tested at 2 048 frames by
    test_kink_flips_explain_the_unfiltered_gradient_excess, where the float64 oracle runs inside the test.)"""
    from hipvae import lib as L
    F, seed = 32768, 23
    arch = ARCHS['vcc']
    gold = np.load(os.path.join(GOLDEN, 'vcc2016_F%d_seed%d_unfiltered.npz' % (F, seed)))
    eng = make_engine('vcc', 'auto', precision=precision)
    exact = precision == 'bf16x3'
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    l3, g = run_train(eng, P, x, y, eps)
    fails, tag = [], 'unfiltered F%d %s ' % (F, precision)
    check(tag + 'loss3', l3, gold['loss3'], TOL_ACT, fails)
    fidx = torch.as_tensor(gold['frame_idx'], device=eng.device)
    for k, width in (('z_mu', 128), ('z_lv', 128), ('xh', 513)):
        got = eng.ws_region(F, L.MODE_TRAIN, k).view(F, width)[fidx].cpu().numpy()
        check(tag + k + ' rows', got, gold[k + '_rows'], TOL_ACT, fails)
    assert np.isfinite(g).all()
    off = 0
    errs, errs32 = [], []
    for i, (name, (poff, shape)) in enumerate(eng.layout.items()):
        n = int(np.prod(shape))
        c = int(gold['grad_sample_counts'][i])
        idx = sample_idx(n, 4096)
        assert idx.size == c
        want = gold['grad_samples'][off:off + c]
        scale = max(float(gold['grad_absmax'][i]), 1e-300)
        e = np.abs(g[poff:poff + n].astype(np.float64)[idx] - want) / scale
        e32 = np.abs(gold['grad_samples_ref32'][off:off + c] - want) / scale
        bar = max(TOL_GRAD, float(e32.max())) if exact else UNFILTERED_X2_TENSOR_BAR
        report(tag + 'grad ' + name + ' (fp32 CPU restatement: %.1e)' % e32.max(), float(e.max()), bar)
        if e.max() > bar:
            fails.append('grad %s %.3e > %.3e' % (name, e.max(), bar))
        errs.append(e)
        errs32.append(e32)
        off += c
    errs, errs32 = np.concatenate(errs), np.concatenate(errs32)
    share, share32 = float((errs > TOL_GRAD).mean()), float((errs32 > TOL_GRAD).mean())
    share_bar = share32 + UNFILTERED_SHARE_SLACK if exact else UNFILTERED_X2_SHARE_BAR
    report(tag + 'share of entries over 2e-4 (fp32 CPU restatement: %.2e)' % share32, share, share_bar)
    if share > share_bar:
        fails.append('share over the bar %.3e > %.3e' % (share, share_bar))
    med = float(np.median(errs))
    report(tag + 'median entry error (fp32 CPU restatement: %.1e)' % np.median(errs32), med, 2e-5)
    if med > 2e-5:
        fails.append('median error %.3e' % med)
    # (5) flips among the near-kink units of the fixture
    ku = np.load(os.path.join(GOLDEN, 'vcc2016_F%d_seed%d_kink_units.npz' % (F, seed)))
    assert float(ku['tau']) == KINK_TAU
    g_ = O.geometry(arch)
    flips = units = near = 0
    for net, layers, pre in (('enc', g_['enc'], 'Encoder/Conv2d-%d/layernorm'), ('dec', g_['dec'][:-1], 'Generator/ConvT-LN%d')):
        for i, l in enumerate(layers):
            k = '%s%d' % (net, i)
            idx = torch.as_tensor(ku[k + '_idx'].astype(np.int64), device=eng.device)
            per = l['cout'] * l['hout']
            a = eng.ws_region(F, L.MODE_TRAIN, '%s_a%d' % (net, i)).view(-1)[idx].cpu().numpy().astype(np.float64)
            st = eng.ws_region(F, L.MODE_TRAIN, '%s_st%d' % (net, i)).view(F, 2).cpu().numpy().astype(np.float64)
            fr = ku[k + '_idx'].astype(np.int64) // per
            ch = (ku[k + '_idx'].astype(np.int64) % per) // l['hout']
            gam = np.asarray(P[(pre % i) + '.scale'], np.float64).ravel()[ch]
            bet = np.asarray(P[(pre % i) + '.offset'], np.float64).ravel()[ch]
            n_gpu = (a - st[fr, 0]) * st[fr, 1] * gam + bet
            n_ref = ku[k + '_n']
            # sanity of the index plumbing: the GPU's n is the float64 n up to the evaluation's own error
            assert np.abs(n_gpu - n_ref).max() < 1e-3, (k, np.abs(n_gpu - n_ref).max())
            fl = int(((n_gpu >= 0) != (n_ref >= 0)).sum())
            flips, units, near = flips + fl, units + int(ku[k + '_units']), near + idx.numel()
    rate = flips / units
    report(tag + 'kink flips: %d of the %d near-kink units (%d units in all); rate' % (flips, near, units), rate, KINK_FLIP_RATE_BAR[precision])
    if rate > KINK_FLIP_RATE_BAR[precision]:
        fails.append('kink flip rate %.2e > %.2e' % (rate, KINK_FLIP_RATE_BAR[precision]))
    assert not fails, '\n'.join(fails)


# lrelu units (LayerNorm outputs) that may land on the other side of the kink than in float64, per evaluated unit: a unit flips
# when |n| is below the evaluation's own error in n (~1e-6 for fp32-exact operands, ~1e-5 for 16-mantissa-bit operand pairs;
# n ~ N(0,1): density 0.4 at 0).  Round 6: the bars sit ~2.5-3x above what was MEASURED at 32 768 frames (78 flips of 5.3e8 units =
# 1.5e-7 with 3-term operands, 1 302 = 2.5e-6 with the default 2 terms; they were 4e-6 / 4e-5, 16x the measurement).  The 2 048-frame
# mechanism test counts ~6 / ~95 flips of 3.8e7 units at the same rates: its bar allows for the Poisson spread of such a small count.
KINK_FLIP_RATE_BAR = {'bf16x3': 5e-7, 'bf16x2': 6e-6}
KINK_FLIP_RATE_BAR_F2048 = {'bf16x3': 1e-6, 'bf16x2': 8e-6}


@pytest.mark.parametrize('precision', ['bf16x3', 'bf16x2'])
def test_kink_flips_explain_the_unfiltered_gradient_excess(precision):
    """An UNFILTERED batch on the large-batch kernels (2 048 plain seeded frames, 38 M lrelu units), float64 oracle run inside
    the test.  What test_unfiltered_benchmark_batch_statistics bounds statistically at 32 768 frames is tested here as a
    mechanism: (1) the units whose branch differs between the GPU and float64 are COUNTED per layer and every one of them lies
    within 1e-4 of the kink in float64 (a flipped unit away from the kink would be an arithmetic error); (2) their rate stays
    under a stated bound per precision; (3) with exactly the near-kink units pinned to the GPU's branch, every gradient tensor
    meets the PLAIN 2e-4 bar in both precisions -- so the excess of the unpinned comparison (reported, not asserted) is the
    flips and nothing else."""
    F, seed = 2048, 29
    arch = ARCHS['vcc']
    eng = make_engine('vcc', 'auto', precision=precision)
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    l3, g = run_train(eng, P, x, y, eps)
    assert np.isfinite(g).all()
    br = gpu_branches(eng, arch, P, F)
    # float64 pre-LN tensors from the PyTorch restatement (agrees with the NumPy one to 4e-15, tests/test_oracle.py; the NumPy
    # loops take a minute at this batch size)
    with torch.no_grad():
        Pt = O.torch_params(P, torch.float64)
        xt, et = torch.tensor(x, dtype=torch.float64), torch.tensor(eps, dtype=torch.float64)
        z_mu, z_lv, eacts = O.torch_encode(arch, Pt, xt)
        _, dacts = O.torch_decode(arch, Pt, z_mu + et * torch.sqrt(torch.exp(z_lv)), torch.tensor(y))
    R = {'enc_a%d' % i: a.numpy()[..., 0] for i, (a, _) in enumerate(eacts)}
    R.update({'dec_a%d' % i: a.numpy()[..., 0] for i, a in enumerate(dacts[1:])})
    fails, tag = [], 'kink F%d %s ' % (F, precision)
    flips = units = 0
    for net, nl, pre in (('enc', 5, 'Encoder/Conv2d-%d/layernorm'), ('dec', 3, 'Generator/ConvT-LN%d')):
        for i in range(nl):
            a = R['%s_a%d' % (net, i)]
            mu, rs = a.mean(axis=(1, 2), keepdims=True), 1 / np.sqrt(a.var(axis=(1, 2), keepdims=True) + 1e-5)
            n = (a - mu) * rs * np.asarray(P[(pre % i) + '.scale'], np.float64).reshape(1, -1, 1) \
                + np.asarray(P[(pre % i) + '.offset'], np.float64).reshape(1, -1, 1)
            fl = (n >= 0) != br['%s%d' % (net, i)]
            far = float(np.abs(n[fl]).max()) if fl.any() else 0.0
            report(tag + '%s%d: %d of %d units flipped; largest |n| among them' % (net, i, int(fl.sum()), n.size), far, KINK_TAU)
            if far >= KINK_TAU:
                fails.append('%s%d: a unit with |n| = %.2e took the other branch' % (net, i, far))
            flips += int(fl.sum())
            units += n.size
    rate = flips / units
    report(tag + 'flip rate (%d of %d units)' % (flips, units), rate, KINK_FLIP_RATE_BAR_F2048[precision])
    if rate > KINK_FLIP_RATE_BAR_F2048[precision]:
        fails.append('flip rate %.2e' % rate)
    _, Gp = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64, kink=br)
    _, Gr = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64)
    worst_raw = 0.0
    for name, (off, shape) in eng.layout.items():
        n = int(np.prod(shape))
        got = g[off:off + n].reshape(shape)
        check(tag + 'pinned grad ' + name, got, Gp[name], TOL_GRAD, fails)
        worst_raw = max(worst_raw, rel_err(got, Gr[name]))
        if name == 'y_embedding/y_emb':
            # the tensor that leads the unfiltered 32 768-frame comparison (2.2e-4 with 2-term operands): the split of its error into
            # operand rounding (flips pinned) and kink flips (the rest of the unpinned figure), measured -- DESIGN.md section 5
            report(tag + 'y_emb: operand rounding alone (flips pinned; reported)', rel_err(got, Gp[name]), float('inf'))
            report(tag + 'y_emb: with the flips (unpinned; reported)', rel_err(got, Gr[name]), float('inf'))
    report(tag + 'UNPINNED worst gradient tensor (reported, not a bar)', worst_raw, float('inf'))
    assert not fails, '\n'.join(fails)


@pytest.mark.parametrize('F,seed,act', [(8192, 21, '0'), (32768, 22, '0'), (8192, 21, '1')])
def test_bf16_mode_against_oracle_fixture(F, seed, act, monkeypatch):
    """BASELINE.json config 2 names bf16: the reduced-precision mode (plain bf16 operands on the GEMM-shaped
    kernels of the bf16 path, fp32 accumulation, fp32 LayerNorm statistics / losses / Adam) is reported beside the
    fp32-class default, never instead of it, and its tolerance is stated separately (3e-2 activations / losses, 6e-2
    gradients) -- at the benchmarked batch size too.  act = '1': with bf16 HBM storage of the thin decoder layers' tensors
    (pre-LN outputs of decoder layers 1 - 2 and the gradients at their activated outputs; VAENPVC_ACT_BF16, off by default
    because it measured slower -- DESIGN.md section 6), same bars."""
    monkeypatch.setenv('VAENPVC_ACT_BF16', act)
    fails = _golden_large(F, seed, 'bf16', 'golden F%d bf16-mode act_bf16=%s ' % (F, act), BF16_TOL_ACT, BF16_TOL_GRAD)
    assert not fails, '\n'.join(fails)


def test_properties_at_benchmarked_batch():
    """Size-independent properties at F = 32768 (the bench batch): the data-parallel identity (gradient of the
    batch == mean of the gradients of its two halves, i.e. what a 2-GPU run computes) and frame-permutation
    equivariance of the conversion path."""
    arch = ARCHS['vcc']
    F = 32768
    eng = make_engine('vcc', 'auto')
    P = O.init_params(arch, 31)
    x, y, eps = O.make_inputs(arch, F, 31)
    l3, g = run_train(eng, P, x, y, eps)
    h = F // 2
    la, ga = run_train(eng, P, x[:h], y[:h], eps[:h])
    lb, gb = run_train(eng, P, x[h:], y[h:], eps[h:])
    assert np.allclose(0.5 * (la + lb), l3, rtol=2e-5)
    for name, (off, shape) in eng.layout.items():
        n = int(np.prod(shape))
        e = rel_err(0.5 * (ga[off:off + n] + gb[off:off + n]), g[off:off + n])
        report('F32768 halves-vs-whole grad ' + name, e, TOL_GRAD)
        assert e < TOL_GRAD, name
    perm = np.random.default_rng(0).permutation(F)
    xt = torch.tensor(x, device=eng.device)
    yt = torch.tensor(y, device=eng.device)
    pt = torch.as_tensor(perm, device=eng.device)
    out = eng.decode(eng.encode(xt), yt)
    outp = eng.decode(eng.encode(xt[pt]), yt[pt])
    assert rel_err(outp.cpu().numpy(), out[pt].cpu().numpy()) < 1e-5


def test_hipgraph_replay_matches_eager():
    """Stepper.capture/replay (one hipGraph launch per train step) follows the eager trajectory."""
    from hipvae.dp import Stepper
    arch = ARCHS['vcc']
    F = 16                                    # the reference's own batch size
    P = O.init_params(arch, 3)
    x, y, eps = O.make_inputs(arch, F, 3)
    res = []
    for use_graph in (False, True):
        eng = make_engine('vcc', 'auto')
        xt, yt, et = upload(eng, P, x, y, eps)
        st = Stepper(eng, 1e-4, 0.5, 0.999)
        g1 = p1 = None
        if use_graph:
            st.capture(xt, yt, et)
            for t in range(3):
                l3 = st.replay()
                if t == 0:
                    g1, p1 = st.grads.clone(), eng.params.clone()
        else:
            for t in range(3):
                l3 = st.step(xt, yt, et)
                if t == 0:
                    g1, p1 = st.grads.clone(), eng.params.clone()
        torch.cuda.synchronize()
        assert st.step_count == 3
        res.append((eng.params.cpu().numpy().copy(), l3.cpu().numpy().copy(), g1.cpu().numpy().copy(), p1.cpu().numpy().copy()))
    p0 = O.flatten_params(P)
    # same kernels, same order; only the summation order of fp32 atomics may differ between the two runs.
    # (1) the FIRST step's gradients (identical parameters) agree to a few ulps of their scale;
    # (2) the first update agrees to ONE max-norm bound wherever the gradient is above the rounding floor (an early
    #     Adam update is ~lr * sign(g): floor-level entries may legitimately flip between ANY two runs);
    # (3) those flips perturb the parameters, so later steps are compared more loosely (max-norm on strong entries,
    #     mean over all): a wrong Adam step on any tensor would be off by ~100 % of the update scale.
    g_e, g_g = res[0][2], res[1][2]
    assert np.abs(g_e - g_g).max() <= 1e-5 * np.abs(g_e).max()
    strong = np.abs(g_e) > 1e-2 * np.abs(g_e).max()
    assert strong.sum() > 1000
    u_e, u_g = res[0][3] - p0, res[1][3] - p0
    assert np.abs(u_e - u_g)[strong].max() <= 1e-3 * np.abs(u_e).max()
    d_eager, d_graph = res[0][0] - p0, res[1][0] - p0
    assert np.abs(d_eager - d_graph)[strong].max() <= 5e-2 * np.abs(d_eager).max()
    assert np.abs(d_eager - d_graph).mean() <= 5e-3 * np.abs(d_eager).max()
    assert np.allclose(res[0][1], res[1][1], rtol=1e-4)


@pytest.mark.parametrize('F', [4096, 4099, 5000, 8209])
@pytest.mark.parametrize('bit,entries,what', [(17, 1025 * 8, 'tap layer'), (16, 7 * 128 * 256, 'encoder layer 4')])
def test_weight_gradient_four_wave_kernels(F, bit, entries, what):
    """From 4 096 frames on the weight gradients of the 1025-tap layer and of encoder layer 4 run on four waves with
    128 x 128 wave tiles, operands by LDS-DMA into a ring of 16-row stages (k_toep_wgrad_bf16_w4, k_gemm_tn4); bits 17 / 16 of
    the backward mask cleared select the eight-wave kernels, which the small sizes pin against the float64 oracle (the
    8 192 / 32 768-frame fixtures hold the default, i.e. the four-wave kernels).  Ragged frame counts exercise the cleared
    tail stage and the requests past the last frame; the two kernels add the same products in another order (bar: 2e-5 of
    the largest entry)."""
    from hipvae import Engine
    arch = ARCHS['vcc']
    eng = Engine(arch)
    P = O.init_params(arch, 3)
    x, y, eps = O.make_inputs(arch, F, 3)
    l_a, g_a = run_train(eng, P, x, y, eps)
    eng.set_tuned_masks(0xffffffff, 0xffffffff & ~(1 << bit))
    l_b, g_b = run_train(eng, P, x, y, eps)
    assert np.isfinite(g_a).all() and np.isfinite(g_b).all()
    (name, (off, shape)), = [(k, v) for k, v in eng.layout.items() if int(np.prod(v[1])) == entries]
    w_a, w_b = g_a[off:off + entries], g_b[off:off + entries]
    worst = np.abs(w_a - w_b).max() / np.abs(w_b).max()
    report('F%d %s weight gradient, four-wave vs eight-wave kernel' % (F, what), worst, 2e-5)
    assert worst < 2e-5, '%s: %.3g of max at flat entry %d' % (name, worst, int(np.abs(w_a - w_b).argmax()))
    rest = np.abs(g_a - g_b)
    rest[off:off + entries] = 0
    assert rest.max() < 1e-5 * np.abs(g_b).max() and rel_err(l_a, l_b) < 1e-6     # nothing else changed (atomic summation order aside)
