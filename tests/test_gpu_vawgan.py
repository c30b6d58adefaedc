"""VAWGAN branch on the GPU against the float64 autograd oracle (oracle/vawgan_oracle.py):
critic step (first and second order terms), the generator-side adversarial gradient through the ConvVAE
backward, and iterations of the trainer's schedule.  The model is a specification (the reference tree holds
only its trainer): PARITY UNPINNED, see the oracle's header."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from helpers import PKG, ROOT, rel_err
from oracle import convvae_oracle as O
from oracle import vawgan_oracle as V
from oracle import philox_ref

pytestmark = pytest.mark.gpu

TOL_VALUE = 1e-4     # losses, critic values
TOL_GRAD = 2e-4      # gradients, relative to the tensor's largest entry (second-order terms included): the ConvVAE bar
                     # (measured worst 4.5e-5, gpurun_out/parity_report.txt)


def worst_ok(tag, worst, tol=None):
    """records the worst per-tensor gradient error of a VAWGAN test in gpurun_out/parity_report.txt (like test_gpu_parity)"""
    import os
    tol = TOL_GRAD if tol is None else tol
    k = max(worst, key=worst.get)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'parity_report.txt')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'a') as fp:
        fp.write('%-70s err=%.3e tol=%.1e %s\n' % ('vawgan ' + tag + ' worst grad ' + k, worst[k], tol, 'OK' if worst[k] < tol else 'FAIL'))
    return worst[k] < tol


def vawgan_arch():
    with open(os.path.join(PKG, 'architecture-vawgan-vcc2016.json')) as fp:
        return json.load(fp)


def critic_inputs(F, seed):
    rng = np.random.RandomState(seed)
    x = np.tanh(rng.randn(F, 513)).astype(np.float32)
    xh = np.tanh(0.7 * rng.randn(F, 513) + 0.2).astype(np.float32)
    t = rng.rand(F).astype(np.float32)
    return x, xh, t


def make_critic(arch, seed):
    from hipvae.critic import Critic
    cr = Critic(arch)
    D = V.disc_init_params(arch, seed)
    assert list(cr.layout.keys()) == list(D.keys())
    for k, (off, shape) in cr.layout.items():
        assert tuple(shape) == D[k].shape, k
    cr.load_flat(V.flatten(D))
    return cr, D


@pytest.mark.parametrize('F', [1, 5, 16])
def test_critic_step_matches_autograd_oracle(F):
    arch = vawgan_arch()
    cr, D = make_critic(arch, 3)
    x, xh, t = critic_inputs(F, 10 + F)
    want, gw = V.critic_loss_and_grads(arch, D, x.astype(np.float64), xh.astype(np.float64), t.astype(np.float64), 10.0)
    dev = cr.device
    grads = torch.full((cr.n_params,), float('nan'), device=dev)
    l2 = cr.critic_fwd_bwd(torch.tensor(x, device=dev), torch.tensor(xh, device=dev), torch.tensor(t, device=dev), 10.0,
                           grads).cpu().numpy()
    vals, _ = cr.values(torch.tensor(x, device=dev), torch.tensor(xh, device=dev))
    vals = vals.cpu().numpy()
    assert rel_err(vals[:F], want['d_real']) < TOL_VALUE and rel_err(vals[F:], want['d_fake']) < TOL_VALUE
    assert abs(l2[0] - want['W_dist']) < TOL_VALUE * max(1.0, abs(want['W_dist']))
    assert abs(l2[1] - want['gp']) < 2e-4 * max(1.0, abs(want['gp']))
    got = cr.param_views(grads.cpu())
    worst = {}
    for k in gw:
        g = got[k].numpy().reshape(gw[k].shape)
        assert np.isfinite(g).all(), k
        worst[k] = float(np.abs(g - gw[k]).max() / max(np.abs(gw[k]).max(), 1e-3))
    assert worst_ok('critic step F%d' % F, worst), worst


def test_critic_and_generator_steps_against_the_golden_fixture():
    """The committed fixture tests/golden/vawgan_F4_seed21.npz (float64 oracle outputs; inputs regenerated from seeds)."""
    import sys as _sys
    _sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import make_golden_vawgan as M
    from helpers import sample_idx
    from hipvae.engine import Engine
    from hipvae.adversarial import AdvStepper
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'vawgan_F4_seed21.npz'))
    arch = vawgan_arch()
    x, y, eps, xh, u = M.inputs(arch)
    P = O.init_params(arch, M.SEED_P)
    cr, D = make_critic(arch, M.SEED_D)
    dev = cr.device
    tt = lambda a, dt=torch.float32: torch.tensor(np.asarray(a), dtype=dt, device=dev)
    grads = torch.empty(cr.n_params, device=dev)
    l2 = cr.critic_fwd_bwd(tt(x), tt(xh), tt(u), M.LAM, grads).cpu().numpy()
    assert abs(l2[0] - gold['critic_losses'][0]) < TOL_VALUE * max(1, abs(gold['critic_losses'][0]))
    assert abs(l2[1] - gold['critic_losses'][1]) < 2e-4 * max(1, gold['critic_losses'][1])
    got = cr.param_views(grads.cpu())
    for i, k in enumerate(cr.layout):
        g = got[k].numpy().astype(np.float64).ravel()
        amax = gold['critic_grad_absmax'][i]
        if amax < 1e-12:
            assert np.abs(g).max() < 1e-6, k          # the dense bias: identically zero
            continue
        assert abs(np.sqrt((g ** 2).sum()) / gold['critic_grad_l2'][i] - 1) < TOL_GRAD, k
        idx = sample_idx(g.size)
        assert np.abs(g[idx] - gold['critic_grad_samples'][i][:len(idx)]).max() < TOL_GRAD * amax, k
        key = 'critic_grad_' + k.replace('/', '__')
        if key in gold.files:
            assert np.abs(g - gold[key].ravel()).max() < TOL_GRAD * amax, k
    eng = Engine(arch, precision='bf16x3')
    eng.load_flat(O.flatten_params(P))
    st = AdvStepper(eng, cr, 1e-4, 0.5, 0.999, M.ALPHA, M.LAM)
    out = st.generator_step(tt(x), tt(y, torch.int64), tt(eps))
    gl = gold['gen_losses']
    assert abs(float(out['D_KL']) / gl[0] - 1) < 1e-4 and abs(float(out['logP']) / gl[1] - 1) < 1e-4
    assert abs(float(out['W_dist']) - gl[2]) < TOL_VALUE * max(1, abs(gl[2]))
    ge, gg = eng.param_views(st.g_e.cpu()), eng.param_views(st.g_g.cpu())
    for i, k in enumerate(eng.layout):
        g = (ge if 'Encoder' in k else gg)[k].numpy().astype(np.float64).ravel()
        amax = gold['gen_grad_absmax'][i]
        assert abs(np.sqrt((g ** 2).sum()) / gold['gen_grad_l2'][i] - 1) < TOL_GRAD, k
        idx = sample_idx(g.size)
        assert np.abs(g[idx] - gold['gen_grad_samples'][i][:len(idx)]).max() < TOL_GRAD * amax, k


def test_critic_step_on_a_generic_geometry():
    """A shrunk critic (54 bins, kernels 5 / 4, 3 and 4 channels: channel counts that divide nothing, even kernel with
    asymmetric SAME padding) through the same kernels."""
    from adv_standin import SMALL_VAWGAN as arch
    cr, D = make_critic(arch, 8)
    rng = np.random.RandomState(5)
    F, H = 9, arch['hwc'][0]
    x, xh, t = np.tanh(rng.randn(F, H)).astype(np.float32), np.tanh(rng.randn(F, H)).astype(np.float32), rng.rand(F).astype(np.float32)
    want, gw = V.critic_loss_and_grads(arch, D, x.astype(np.float64), xh.astype(np.float64), t.astype(np.float64), 10.0)
    dev = cr.device
    grads = torch.full((cr.n_params,), float('nan'), device=dev)
    l2 = cr.critic_fwd_bwd(torch.tensor(x, device=dev), torch.tensor(xh, device=dev), torch.tensor(t, device=dev), 10.0,
                           grads).cpu().numpy()
    assert abs(l2[0] - want['W_dist']) < TOL_VALUE * max(1.0, abs(want['W_dist'])) and abs(l2[1] - want['gp']) < 2e-4 * max(1.0, want['gp'])
    got = cr.param_views(grads.cpu())
    worst = {k: float(np.abs(got[k].numpy().reshape(gw[k].shape) - gw[k]).max() / max(np.abs(gw[k]).max(), 1e-3)) for k in gw}
    assert worst_ok('critic step generic geometry', worst), worst
    target, _ = cr.generator_target(torch.tensor(x, device=dev), torch.tensor(xh, device=dev), 50.0)
    import torch as T
    Dt = O.torch_params(D, T.float64)
    xht = T.tensor(xh.astype(np.float64), requires_grad=True)
    g, = T.autograd.grad(V.torch_discriminate(arch, Dt, xht).sum(), xht)
    assert rel_err(target.cpu().numpy() - x, 50.0 * (1 + 1e-6) * g.numpy()) < TOL_VALUE


def assert_repeatable(cr, ga, gb):
    """Two runs of the same critic step: bitwise equal for every tensor whose gradient is accumulated by one thread per
    element (the 115-tap layer); the two thin conv layers' gradients come from the job-list launch of csrc/disc_frame.h,
    whose frame chunks meet through fp32 atomics, the dense unit's own from atomics inside its forward kernel: equal up to
    the order of a few additions."""
    for k, (off, shp) in cr.layout.items():
        n = int(np.prod(shp))
        a, b = ga[off:off + n], gb[off:off + n]
        if 'Conv2d-0' in k or 'Conv2d-1' in k or '/dense/' in k:      # (the dense unit's own gradient: atomics inside its forward kernel)
            assert float((a - b).abs().max()) <= 2e-6 * max(float(a.abs().max()), 1e-12), k
        else:
            assert torch.equal(a, b), k


def test_large_batch_dense_layer_agrees_with_the_conv_kernels(monkeypatch):
    """F = 1024 (3072 rows through every critic kernel: many GEMM tiles, frame-split weight gradients): finite, bitwise
    repeatable, and the 115-tap layer as a dense layer on the matrix cores == the same layer on the thread-per-output
    conv kernels (VAENPVC_DISC_DENSE=0, read when the critic is created) up to fp32 summation order."""
    from hipvae.critic import Critic
    arch = vawgan_arch()
    F = 1024
    cr, _ = make_critic(arch, 6)
    g = torch.Generator().manual_seed(3)
    dev = cr.device
    x = torch.tanh(torch.randn(F, 513, generator=g)).to(dev)
    xh = torch.tanh(0.7 * torch.randn(F, 513, generator=g) + 0.2).to(dev)
    t = torch.rand(F, generator=g).to(dev)
    g1, g2, g3 = (torch.empty(cr.n_params, device=dev) for _ in range(3))
    l_a = cr.critic_fwd_bwd(x, xh, t, 10.0, g1).clone()
    l_b = cr.critic_fwd_bwd(x, xh, t, 10.0, g2).clone()
    assert torch.isfinite(g1).all() and torch.equal(l_a, l_b)
    assert_repeatable(cr, g1, g2)
    monkeypatch.setenv('VAENPVC_DISC_DENSE', '0')
    c2 = Critic(arch)
    monkeypatch.delenv('VAENPVC_DISC_DENSE')
    c2.params.copy_(cr.params)
    l_c = c2.critic_fwd_bwd(x, xh, t, 10.0, g3).clone()
    assert torch.allclose(l_a, l_c, rtol=1e-5, atol=1e-6)
    for k, (off, shp) in cr.layout.items():
        n = int(np.prod(shp))
        a, b = g1[off:off + n], g3[off:off + n]
        assert float((a - b).abs().max()) <= 3e-4 * max(float(b.abs().max()), 1e-6), k
    tgt1, _ = cr.generator_target(x, xh, 50.0)
    tgt2, _ = c2.generator_target(x, xh, 50.0)
    assert float((tgt1 - tgt2).abs().max()) <= 1e-4 * float((tgt2 - x).abs().max())


@pytest.mark.parametrize('F', [5, 16, 300])
def test_front_kernels_agree_with_the_per_layer_kernels(monkeypatch, F):
    """The two thin conv layers on the per-row kernels + job-list weight gradient of csrc/disc_frame.h (default for the
    VCC2016 layer table) against the same layers on the per-layer kernels (VAENPVC_DISC_FRONT=0, read when the critic is
    created): critic step (losses, every gradient tensor) and generator target, up to fp32 summation order.  F = 300:
    more rows (900) than workgroups, frame chunks in the job list."""
    from hipvae.critic import Critic
    arch = vawgan_arch()
    cr, _ = make_critic(arch, 9)
    monkeypatch.setenv('VAENPVC_DISC_FRONT', '0')
    c2 = Critic(arch)
    monkeypatch.delenv('VAENPVC_DISC_FRONT')
    c2.params.copy_(cr.params)
    g = torch.Generator().manual_seed(F)
    dev = cr.device
    x = torch.tanh(torch.randn(F, 513, generator=g)).to(dev)
    xh = torch.tanh(0.7 * torch.randn(F, 513, generator=g) + 0.2).to(dev)
    t = torch.rand(F, generator=g).to(dev)
    g1, g2 = (torch.empty(cr.n_params, device=dev) for _ in range(2))
    l_a = cr.critic_fwd_bwd(x, xh, t, 10.0, g1).clone()
    l_b = c2.critic_fwd_bwd(x, xh, t, 10.0, g2).clone()
    assert torch.allclose(l_a, l_b, rtol=1e-5, atol=1e-6)
    for k, (off, shp) in cr.layout.items():
        n = int(np.prod(shp))
        a, b = g1[off:off + n], g2[off:off + n]
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-6), k
    tgt1, _ = cr.generator_target(x, xh, 50.0)
    tgt2, _ = c2.generator_target(x, xh, 50.0)
    assert float((tgt1 - tgt2).abs().max()) <= 1e-5 * float((tgt2 - x).abs().max())


def test_critic_step_is_deterministic_and_linear_in_lambda():
    """Repeatable (see assert_repeatable), and grad(lambda) is affine in lambda: g(20) - g(10) == g(10) - g(0)."""
    arch = vawgan_arch()
    cr, _ = make_critic(arch, 4)
    x, xh, t = (torch.tensor(a, device=cr.device) for a in critic_inputs(7, 2))
    g = [torch.empty(cr.n_params, device=cr.device) for _ in range(4)]
    for gi, lam in zip(g, (0.0, 10.0, 20.0, 10.0)):
        cr.critic_fwd_bwd(x, xh, t, lam, gi)
    assert_repeatable(cr, g[1], g[3])
    d1, d2 = (g[2] - g[1]).cpu().numpy(), (g[1] - g[0]).cpu().numpy()
    assert np.abs(d1 - d2).max() < 1e-4 * np.abs(d2).max()


def test_generator_step_gradients_match_oracle():
    """l_E on 'Encoder', l_G = -logP + alpha W_dist on 'Generator' + 'y_emb', from the two ConvVAE passes."""
    from hipvae.engine import Engine
    from hipvae.adversarial import AdvStepper, name_ranges
    arch = vawgan_arch()
    F, alpha = 16, 50.0
    eng = Engine(arch, precision='bf16x3')
    P = O.init_params(arch, 7)
    eng.load_flat(O.flatten_params(P))
    cr, D = make_critic(arch, 5)
    x, y, eps = O.make_inputs(arch, F, 9)
    want, gw = V.encoder_generator_grads(arch, P, D, x, y, eps, alpha)
    st = AdvStepper(eng, cr, 1e-4, 0.5, 0.999, alpha, 10.0)
    p0 = eng.params.clone()
    dev = eng.device
    out = st.generator_step(torch.tensor(x, dtype=torch.float32, device=dev), torch.tensor(y, device=dev),
                            torch.tensor(eps, dtype=torch.float32, device=dev))
    assert abs(float(out['W_dist']) - want['W_dist']) < TOL_VALUE * max(1.0, abs(want['W_dist']))
    assert abs(float(out['logP']) - want['logP']) < 1e-4 * abs(want['logP'])
    assert abs(float(out['D_KL']) - want['D_KL']) < 1e-4 * abs(want['D_KL'])
    ge, gg = eng.param_views(st.g_e.cpu()), eng.param_views(st.g_g.cpu())
    worst = {}
    for k, g in gw.items():
        got = (ge if 'Encoder' in k else gg)[k].numpy().reshape(g.shape)
        worst[k] = float(np.abs(got - g).max() / max(np.abs(g).max(), 1e-6))
    assert worst_ok('generator step', worst), worst
    # the applies touched exactly the three groups, Encoder with t = 1 and Generator / y_emb with t = 2
    assert st.applies == 2 and st.step_count == 1
    changed = (eng.params != p0).cpu().numpy()
    cover = np.zeros(eng.n_params, bool)
    for lo, hi in name_ranges(eng.layout, lambda n: True):
        cover[lo:hi] = True
    assert cover.all() and changed.mean() > 0.99
    pv, p0v = eng.param_views(eng.params.cpu()), eng.param_views(p0.cpu())
    for k, t in (('Encoder/Conv2d-2/kernel', 1), ('Generator/conv2d_transpose_1/kernel', 2), ('y_embedding/y_emb', 2)):
        wantp, _, _ = O.tf_adam_step(P[k], gw[k], 0.0, 0.0, t, 1e-4, 0.5, 0.999)
        got_delta = (pv[k] - p0v[k]).numpy().reshape(P[k].shape)
        assert rel_err(got_delta, wantp - P[k]) < 2e-3, k


def test_two_iterations_follow_the_oracle_trajectory():
    """trainer/vae.py:176-179 with nIterD = 2: critic, critic, generator (encoder then generator apply), twice;
    every step on its own batch; one Adam apply counter across the three groups."""
    from hipvae.engine import Engine
    from hipvae.adversarial import AdvStepper
    arch = vawgan_arch()
    F, n_d, iters = 8, 2, 2
    lr, b1, b2, alpha, lam = 1e-4, 0.5, 0.999, 50.0, 10.0
    eng = Engine(arch, precision='bf16x3')
    P = O.init_params(arch, 11)
    eng.load_flat(O.flatten_params(P))
    cr, D = make_critic(arch, 12)
    batches = []
    for i in range(iters * (n_d + 1)):
        x, y, eps = O.make_inputs(arch, F, 100 + i)
        batches.append(dict(x=x, y=y, eps=eps, u=np.random.RandomState(200 + i).rand(F)))
    Pw, Dw, log, strong = V.train_iterations(arch, P, D, batches, lr, b1, b2, alpha, lam, n_d)
    st = AdvStepper(eng, cr, lr, b1, b2, alpha, lam)
    dev = eng.device
    tt = lambda a, dt=torch.float32: torch.tensor(np.asarray(a), dtype=dt, device=dev)
    it = iter(batches)
    for _ in range(iters):
        bs = [next(it) for _ in range(n_d)]     # one generator forward for the n_d critic batches
        st.critic_steps([(tt(b['x']), tt(b['y'], torch.int64)) for b in bs], [tt(b['eps']) for b in bs], [tt(b['u']) for b in bs])
        b = next(it)
        out = st.generator_step(tt(b['x']), tt(b['y'], torch.int64), tt(b['eps']))
    assert st.applies == iters * (n_d + 2) and st.step_count == iters
    assert abs(float(out['W_dist']) - log[-1]['W_dist']) < 5e-3 * max(1.0, abs(log[-1]['W_dist']))
    # Compare the parameter MOVES.  Adam divides every entry by its own gradient history (and here starts at t = 3,
    # where sqrt(v) is only ~0.03 |g| against eps = 1e-8), so entries near a float32 implementation's noise floor
    # follow no particular trajectory: the tight bar applies where the gradient was well above it in every apply
    # (as in tests/test_gpu_plugins.py), a loose statistical bar everywhere.
    dv, pv = cr.param_views(cr.params.cpu()), eng.param_views(eng.params.cpu())
    n_strong = 0
    for got, want, start in ((dv, Dw, D), (pv, Pw, P)):
        for k in want:
            move = np.abs(want[k] - start[k]).max()
            if move < 1e-9:                              # the critic's dense bias: its gradient is identically zero
                continue
            dev = np.abs(got[k].numpy().reshape(want[k].shape) - want[k])
            ok = strong[k]
            n_strong += int(ok.sum())
            if ok.any():
                assert dev[ok].mean() < 0.03 * move and np.quantile(dev[ok], 0.99) < 0.15 * move, (k, dev[ok].mean() / move)
                assert dev[ok].max() < 1.1 * move, (k, dev[ok].max() / move)   # (m = 0.25 g1 + 0.5 g2 can cancel)
            assert dev.mean() < 0.08 * move, (k, dev.mean() / move)
    assert n_strong > 20000


@pytest.mark.parametrize('F,precision', [(16, 'bf16x2'), (256, 'bf16x2'), (1027, 'bf16x2'), (2048, 'bf16'), (2048, 'bf16x3')])
def test_backward_only_pass_equals_the_full_step(F, precision):
    """vaenpvc_train_bwd_target on the activations of a preceding step == vaenpvc_train_fwd_bwd_target (the backward
    pass must not have overwritten anything a second backward reads), at batch sizes of every kernel family."""
    from hipvae.engine import Engine
    arch = vawgan_arch()
    eng = Engine(arch, precision=precision)
    eng.init_params(3)
    dev = eng.device
    g = torch.Generator(device='cpu').manual_seed(F)
    x = torch.tanh(torch.randn(F, 513, generator=g)).to(dev)
    y = torch.randint(0, 10, (F,), generator=g).to(dev)
    eps = torch.randn(F, 128, generator=g).to(dev)
    target = (x + 0.3 * torch.randn(F, 513, generator=g).to(dev)).contiguous()
    g0, g1, g2 = (torch.empty(eng.n_params, device=dev) for _ in range(3))
    l3 = torch.zeros(3, device=dev)
    eng.train_fwd_bwd_target(x, y, eps, target, g1, out=l3)
    want_l3 = l3.clone()
    eng.train_fwd_bwd(x, y, eps, g0)
    eng.train_bwd_target(x, y, eps, target, g2, out=l3)
    assert torch.allclose(l3, want_l3, rtol=1e-5, atol=0)      # (two forward passes: equal up to the split-K summation order)
    scale = g1.abs().max()
    # (several weight gradients accumulate with atomics -- split-K, frame chunks -- so runs differ by rounding; a
    #  tensor the backward pass had overwritten would be off by O(1))
    assert (g2 - g1).abs().max() <= 5e-5 * scale, float((g2 - g1).abs().max() / scale)
    assert (g0 - g1).abs().max() > 1e-3 * scale          # and the target does matter
    eng.train_bwd_target(x, y, eps, x, g2, out=l3)        # target = x: the plain step's gradient again
    assert (g2 - g0).abs().max() <= 5e-5 * g0.abs().max(), float((g2 - g0).abs().max() / g0.abs().max())


def test_uniform_draw_matches_numpy_philox():
    from hipvae.engine import Engine
    eng = Engine(vawgan_arch())
    got = eng.philox_uniform(1000, 77, 5).cpu().numpy()
    assert np.array_equal(got, philox_ref.uniform(1000, 77, 5))
    assert got.min() >= 0.0 and got.max() < 1.0


def test_vawgan_plugins_train_and_checkpoint(tmp_path):
    """main.py wiring with --model VAWGAN --trainer VAWGANTrainer on synthetic .bin data: loss keys, schedule
    (nIterD critic batches + 1 generator batch per iteration), status line, checkpoint round trip."""
    import analyzer
    from model.vawgan import VAWGAN
    from trainer.vae import VAWGANTrainer
    from test_gpu_plugins import make_dataset
    arch = vawgan_arch()
    arch['training'].update(max_iter=2, batch_size=16, nIterD=3)
    allr, xmin, xmax = make_dataset(str(tmp_path))
    arch['training']['datadir'] = [os.path.join(str(tmp_path), 'bin', 'Training Set', s, '*.bin') for s in ('SF1', 'TM3')]
    image, label = analyzer.read(arch['training']['datadir'], 16, normalizer=analyzer.Tanhize(xmax=xmax, xmin=xmin), seed=3)
    machine = VAWGAN(arch, seed=5)
    n_batches = [0]
    orig = image.source.next_batch
    def spy():
        n_batches[0] += 1
        return orig()
    image.source.next_batch = spy
    loss = machine.loss(image, label)
    assert set(loss.keys()) == {'l_D', 'l_E', 'l_G', 'D_KL', 'logP', 'W_dist', 'gp'}
    dirs = {'logdir': os.path.join(str(tmp_path), 'logdir', 'train', 'stamp')}
    trainer = VAWGANTrainer(loss, arch, types.SimpleNamespace(seed=17, restore_from=None, ckpt=None), dirs)
    assert set(trainer.opt.keys()) == {'d', 'g', 'e', 'global_step'}
    d0, p0 = machine.critic.params.clone(), machine.engine.params.clone()
    path = trainer.train(nIter=2)
    st = trainer.opt['g']
    assert n_batches[0] == 2 * 4 and st.step_count == 2 and st.applies == 2 * 5
    assert os.path.basename(path) == 'model.ckpt-2'
    assert not torch.equal(d0, machine.critic.params) and not torch.equal(p0, machine.engine.params)
    msg = trainer._refresh_status()
    assert msg.startswith('Iter 00002: W_dist = ') and 'GP = ' in msg and 'D_KL(z) = ' in msg
    assert all(np.isfinite(float(v)) for v in st.status.values())
    # eager loss on tensors
    x, y = orig()
    ev = machine.loss(x, y)
    assert abs(float(ev['l_D']) - (-float(ev['W_dist']) + 10 * float(ev['gp']))) < 1e-3 * (1 + abs(float(ev['l_D'])))
    # restore into a fresh machine
    m2 = VAWGAN(arch, seed=99)
    t2 = VAWGANTrainer(m2.loss(image, label), arch, types.SimpleNamespace(seed=17, restore_from=None, ckpt=None),
                       {'logdir': dirs['logdir']})
    assert t2.restore(dirs['logdir']) == 2
    assert torch.equal(m2.critic.params, machine.critic.params) and torch.equal(m2.engine.params, machine.engine.params)
    assert t2.opt['g'].applies == st.applies
    # a VAWGAN checkpoint converts like a ConvVAE one (convert.py:79-89 loads the model by name and restores it)
    from util.wrapper import load
    import model.vae
    m3 = model.vae.VAWGAN(arch, seed=7)                  # `--model VAWGAN` with the default --model_module
    assert load(m3.engine, dirs['logdir']) == 2
    z = machine.encode(x)
    # (small batches split the heads' K over workgroups with atomics: equal up to summation order)
    assert torch.allclose(m3.encode(x), z, rtol=0, atol=2e-5) and torch.allclose(m3.decode(z, y), machine.decode(z, y), rtol=0, atol=2e-5)


def test_critic_argument_errors():
    import ctypes as C
    from hipvae import HipVaeError
    arch = vawgan_arch()
    cr, _ = make_critic(arch, 1)
    dev = cr.device
    x = torch.zeros(4, 513, device=dev)
    g = torch.zeros(cr.n_params, device=dev)
    with pytest.raises(TypeError):
        cr.critic_fwd_bwd(x, x, torch.zeros(3, device=dev), 10.0, g)
    with pytest.raises(ValueError):
        cr.critic_fwd_bwd(x, torch.zeros(5, 513, device=dev), torch.zeros(4, device=dev), 10.0, g)
    with pytest.raises(TypeError):
        cr.critic_fwd_bwd(x, x, torch.zeros(4, device=dev), 10.0, g[:-1])
    ws = torch.zeros(1024, dtype=torch.uint8, device=dev)
    l2 = torch.zeros(2, device=dev)
    t = torch.zeros(4, device=dev)
    rc = cr.lib.vaenpvc_disc_critic_fwd_bwd(cr.handle, cr.params.data_ptr(), x.data_ptr(), x.data_ptr(), t.data_ptr(), 4, 10.0,
                                            g.data_ptr(), l2.data_ptr(), ws.data_ptr(), 1024, None)
    assert rc == -2 and b'workspace too small' in cr.lib.vaenpvc_last_error()
    rc = cr.lib.vaenpvc_disc_critic_fwd_bwd(cr.handle, cr.params.data_ptr(), x.data_ptr(), None, t.data_ptr(), 4, 10.0,
                                            g.data_ptr(), l2.data_ptr(), ws.data_ptr(), 1024, None)
    assert rc == -1 and b'null' in cr.lib.vaenpvc_last_error()
    bad = dict(arch)
    bad['discriminator'] = dict(arch['discriminator'], output=[16, 32, 300])
    from hipvae.critic import Critic
    with pytest.raises(HipVaeError):
        Critic(bad)                                       # more than 256 channels: VAENPVC_E_UNSUPPORTED, not a wrong result


RCCL_ADV = r"""
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, 'vae-npvc_amd')); sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29643', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from hipvae import Engine
from hipvae.critic import Critic
from hipvae.adversarial import AdvStepper
arch = json.load(open(os.path.join(%(root)r, 'vae-npvc_amd', 'architecture-vawgan-vcc2016.json')))
g = torch.Generator().manual_seed(1)
x = torch.tanh(torch.randn(16, 513, generator=g)).cuda(); y = torch.randint(0, 10, (16,), generator=g).cuda()
eps = torch.randn(16, 128, generator=g).cuda(); t = torch.rand(16, generator=g).cuda()
out = {}
for mode in ('plain', 'rccl'):
    os.environ['VAENPVC_FORCE_DIST'] = '1' if mode == 'rccl' else '0'
    eng, cr = Engine(arch), Critic(arch)
    eng.init_params(2); cr.init_params(3)
    st = AdvStepper(eng, cr, 1e-4, 0.5, 0.999, 50.0, 10.0)
    assert st.collective == (mode == 'rccl')
    st.broadcast_params()
    for _ in range(2):
        l2 = st.critic_step(x, y, eps, t)
    l = st.generator_step(x, y, eps)
    torch.cuda.synchronize()
    out[mode] = (eng.params.cpu().numpy(), cr.params.cpu().numpy(), st.g_d.cpu().numpy(), st.g_g.cpu().numpy(),
                 np.array([float(l2[0]), float(l2[1]), float(l['W_dist']), float(l['logP'])]))
a, b = out['plain'], out['rccl']
for i in (2, 3):      # gradients of the last steps: equal up to the order of fp32 atomics / earlier Adam moves
    assert np.abs(a[i] - b[i]).max() <= 1e-3 * np.abs(a[i]).max(), i
assert np.allclose(a[4], b[4], rtol=1e-3), (a[4], b[4])
assert np.abs(a[0] - b[0]).mean() <= 2e-5 and np.abs(a[1] - b[1]).mean() <= 2e-5
dist.destroy_process_group()
print('RCCL_ADV_OK')
"""


def test_single_rank_rccl_adversarial_steps():
    """hipvae.adversarial over the real "nccl" (RCCL) backend with one rank (VAENPVC_FORCE_DIST=1): the all-reduces of the
    critic and of both ConvVAE gradient buffers, the broadcast of both parameter sets and the loss means run through
    RCCL and follow the collective-free trajectory."""
    r = subprocess.run([sys.executable, '-c', RCCL_ADV % {'root': ROOT}], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'RCCL_ADV_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
