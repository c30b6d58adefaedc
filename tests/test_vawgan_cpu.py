"""CPU checks of the VAWGAN branch: the oracle against finite differences and its own closed
forms, the critic's host-side table through the C-ABI, and the adversarial stepper's host logic (variable groups,
shared Adam apply counter, shifted-target generator gradient, 2-rank gloo) on oracle-backed stand-ins."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from adv_standin import SMALL_VAWGAN, EngineStandIn, CriticStandIn, adv_batches
from helpers import PKG
from oracle import convvae_oracle as O
from oracle import vawgan_oracle as V

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dp_gloo_worker.py')


def vcc_arch():
    with open(os.path.join(PKG, 'architecture-vawgan-vcc2016.json')) as fp:
        return json.load(fp)


def test_architecture_file_and_geometry():
    arch = vcc_arch()
    assert arch['training']['nIterD'] == 5 and arch['training']['lambda'] == 10 and arch['training']['alpha'] == 50.0
    g = V.disc_geometry(arch)
    assert [(l['cin'], l['hin'], l['cout'], l['hout'], l['k'], l['pad']) for l in g] == \
        [(1, 513, 16, 171, 7, 2), (16, 171, 32, 57, 7, 2), (32, 57, 64, 19, 115, 56)]
    assert sum(int(np.prod(s)) for s in V.disc_param_layout(arch).values()) == 240769


def test_critic_loss_gradient_against_finite_differences():
    """l_D = -W_dist + lambda gp: autograd (with the second-order term) vs central differences in float64."""
    arch = SMALL_VAWGAN
    D = V.disc_init_params(arch, 1)
    rng = np.random.RandomState(0)
    F, H = 3, arch['hwc'][0]
    x, xh, u = np.tanh(rng.randn(F, H)), np.tanh(rng.randn(F, H)), rng.rand(F)
    out, g = V.critic_loss_and_grads(arch, D, x, xh, u, 10.0)
    assert out['gp'] > 1e-3          # the penalty is active, so its gradient is exercised

    def l_D(Dn):
        T = V.torch_critic_terms(arch, O.torch_params(Dn, torch.float64), torch.tensor(x), torch.tensor(xh),
                                 torch.tensor(u), False)
        return float(-T['W_dist'] + 10.0 * T['gp'])
    for k in D:
        flat = D[k].reshape(-1)
        for i in rng.choice(flat.size, size=min(3, flat.size), replace=False):
            D2 = {a: b.copy() for a, b in D.items()}
            h = 1e-6
            D2[k].reshape(-1)[i] = flat[i] + h
            up = l_D(D2)
            D2[k].reshape(-1)[i] = flat[i] - h
            dn = l_D(D2)
            fd = (up - dn) / (2 * h)
            assert abs(fd - g[k].reshape(-1)[i]) < 1e-6 * max(1.0, abs(fd)), (k, i, fd, g[k].reshape(-1)[i])


def test_layernorm_double_backward_closed_form():
    """The closed form csrc/disc.hip:k_ln_bwd_bwd implements == autograd of the LayerNorm input gradient."""
    rng = np.random.RandomState(3)
    for N in (7, 50, 2736):
        u = torch.tensor(rng.randn(N), requires_grad=True)
        p = torch.tensor(rng.randn(N), requires_grad=True)
        q = torch.tensor(rng.randn(N))
        mu = u.mean()
        r = torch.rsqrt(((u - mu) ** 2).mean() + O.LN_EPS)
        xhat = (u - mu) * r
        ub = r * (p - p.mean() - xhat * (p * xhat).mean())
        assert np.allclose(ub.detach().numpy(), V.np_ln_bwd(p.detach().numpy(), xhat.detach().numpy(), float(r.detach())))
        gp_, gu_ = torch.autograd.grad((ub * q).sum(), [p, u])
        pt, ut = V.np_ln_bwd_bwd(q.numpy(), p.detach().numpy(), xhat.detach().numpy(), float(r.detach()))
        assert np.abs(pt - gp_.numpy()).max() < 1e-12 and np.abs(ut - gu_.numpy()).max() < 1e-11


@pytest.mark.parametrize('which', ['vcc', 'small'])
def test_critic_table_through_the_abi_matches_oracle_layout(which):
    from hipvae import lib as L
    from hipvae.critic import disc_arch_to_struct
    arch = vcc_arch() if which == 'vcc' else SMALL_VAWGAN
    lib = L.load_library()
    a = disc_arch_to_struct(arch)
    h = C.c_void_p()
    assert lib.vaenpvc_disc_create(C.byref(a), C.byref(h)) == 0
    want = V.disc_param_layout(arch)
    assert lib.vaenpvc_disc_param_count(h) == len(want)
    buf = C.create_string_buffer(128)
    off, nd, shp = C.c_int64(), C.c_int32(), (C.c_int64 * 4)()
    pos = 0
    for i, (name, shape) in enumerate(want.items()):
        assert lib.vaenpvc_disc_param_info(h, i, buf, 128, C.byref(off), C.byref(nd), shp) == 0
        assert buf.value.decode() == name and tuple(shp[k] for k in range(nd.value)) == tuple(shape) and off.value == pos
        pos += int(np.prod(shape))
    assert lib.vaenpvc_disc_param_floats(h) == pos
    b16, b32 = lib.vaenpvc_disc_workspace_bytes(h, 16), lib.vaenpvc_disc_workspace_bytes(h, 32)
    assert 0 < b16 < b32 <= 2 * b16 + 4096
    assert lib.vaenpvc_disc_workspace_bytes(h, 0) < 0 and b'F' in lib.vaenpvc_last_error()
    assert lib.vaenpvc_disc_param_info(h, len(want), buf, 128, None, None, None) < 0
    lib.vaenpvc_disc_destroy(h)
    a.n_layers = 0
    assert lib.vaenpvc_disc_create(C.byref(a), C.byref(h)) < 0


def test_variable_groups_are_contiguous_ranges_of_the_table():
    """trainer/vae.py:128-130 groups by name; in the flat table 'Encoder' is one range, 'Generator' + 'y_emb' two."""
    from hipvae.adversarial import name_ranges
    from hipvae import lib as L
    from hipvae.engine import arch_to_struct
    lib = L.load_library()
    ctx = C.c_void_p()
    a = arch_to_struct(vcc_arch())
    assert lib.vaenpvc_ctx_create(C.byref(a), C.byref(ctx)) == 0
    lay, off = {}, 0
    for n, shp in O.param_layout(vcc_arch()).items():
        lay[n] = (off, shp)
        off += int(np.prod(shp))
    enc = name_ranges(lay, lambda n: 'Encoder' in n)
    gen = name_ranges(lay, lambda n: 'Generator' in n or 'y_emb' in n)
    assert len(enc) == 1 and len(gen) == 2 and gen[0] == (0, 1280) and gen[1][0] == enc[0][1] and gen[1][1] == off
    assert enc[0][0] == 1280 and sum(hi - lo for lo, hi in enc + gen) == off == lib.vaenpvc_param_floats(ctx)
    lib.vaenpvc_ctx_destroy(ctx)


def test_stepper_iteration_matches_direct_autograd_trajectory():
    """AdvStepper on the oracle-backed stand-ins == oracle.train_iterations (direct autograd of l_D / l_E / l_G):
    pins the shifted-target formulation of the generator gradient, the groups and the shared apply counter."""
    from hipvae.adversarial import AdvStepper
    arch = SMALL_VAWGAN
    F, n_d, iters = 4, 2, 2
    be, cr = EngineStandIn(arch, 10), CriticStandIn(arch, 11)
    P0 = O.unflatten_params(arch, be.params.numpy().copy())
    D0 = cr._D()
    batches = adv_batches(arch, F, iters * (n_d + 1), 60)
    Pw, Dw, log, _ = V.train_iterations(arch, P0, D0, batches, 1e-3, 0.5, 0.999, 50.0, 10.0, n_d)
    st = AdvStepper(be, cr, 1e-3, 0.5, 0.999, 50.0, 10.0)
    it = iter(batches)
    tt = torch.tensor
    for _ in range(iters):
        bs = [next(it) for _ in range(n_d)]     # one generator forward for the n_d critic batches
        l2 = st.critic_steps([(tt(b['x']), tt(b['y'])) for b in bs], [tt(b['eps']) for b in bs], [tt(b['u']) for b in bs])
        b = next(it)
        out = st.generator_step(tt(b['x']), tt(b['y']), tt(b['eps']))
    assert st.applies == iters * (n_d + 2) and st.step_count == iters
    # (the stepper keeps its loss vectors in float32)
    assert abs(float(out['W_dist']) - log[-1]['W_dist']) < 1e-6 and abs(float(out['logP']) / log[-1]['logP'] - 1) < 1e-6
    assert np.abs(be.params.numpy() - np.concatenate([Pw[k].ravel() for k in Pw])).max() < 1e-9
    assert np.abs(cr.params.numpy() - V.flatten(Dw)).max() < 1e-9
    assert float(l2[1]) >= 0 and set(st.status) == {'D_KL', 'logP', 'W_dist', 'gp'}


def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path):
    port = 31000 + os.getpid() % 2000
    res = {}
    for world in (2, 1):
        out = str(tmp_path / ('adv%d.npy' % world))
        procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port + world), '8', '2', out, 'adv'])
                 for r in range(world)]
        try:
            for p in procs:
                assert p.wait(timeout=240) == 0
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        res[world] = np.load(out)
    move = np.abs(res[1][:-4]).max()
    assert np.abs(res[2][:-4] - res[1][:-4]).max() < 1e-6 * move
    assert np.allclose(res[2][-4:], res[1][-4:], rtol=1e-8)      # D_KL, logP, W_dist, gp: means over ranks == global


@pytest.mark.parametrize('world', [1, 2])
def test_vawgan_trainer_loop_schedule_status_and_restore(tmp_path, world):
    """VAWGANTrainer.train on the stand-ins (tests/dp_gloo_worker.py:run_adv_trainer): batches consumed per iteration,
    status line, checkpoint, restore-and-continue; with two ranks the status path must not issue collectives of its own."""
    port = 33000 + os.getpid() % 2000 + world
    out = str(tmp_path / 'advt.npy')
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), '8', '2', out, 'adv_trainer'])
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert os.path.exists(out)


def test_golden_fixture_reproduced_by_the_oracle():
    """tests/golden/vawgan_F4_seed21.npz (made by tests/golden/make_golden_vawgan.py) pins the specification against
    accidental changes of the oracle; the GPU suite compares the HIP path with the same file."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_golden_vawgan as M
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vawgan_F4_seed21.npz'))
    now = M.run()
    assert set(now.keys()) == set(gold.files)
    for k in gold.files:
        assert np.allclose(now[k], gold[k], rtol=1e-9, atol=1e-12), k
