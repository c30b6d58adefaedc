"""One rank of the world_size-N gloo data-parallel tests (launched by test_dp_gloo.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, 'vae-npvc_amd')):
    sys.path.insert(0, p)
from helpers import SMALL_ARCH  # noqa: E402
from oracle import convvae_oracle as O  # noqa: E402
from oracle import philox_ref  # noqa: E402


class OracleBackend(object):
    """CPU stand-in for hipvae.Engine built from the oracle (test infrastructure).  It mirrors the engine's
    contract towards hipvae.dp: losses are written into `out`, the sampler draw comes from (seed, offset)
    when eps is None, and the flat gradient buffer is reported back to front in the library's four
    contiguous ranges (decoder convs, merge, heads, embedding + encoder) through the bucket callback."""

    def __init__(self, arch, seed):
        self.arch = arch
        lay = O.param_layout(arch)
        self.names = list(lay.keys())
        self.params = torch.tensor(O.flatten_params(O.init_params(arch, seed)), dtype=torch.float64)
        self.n_params = self.params.numel()
        off, offs = 0, {}
        for n, shp in lay.items():
            offs[n] = off
            off += int(np.prod(shp))
        cut = [offs['Generator/conv2d_transpose/kernel'], offs['Generator/fully_connected/weights'],
               offs['Encoder/dense/kernel'], 0]
        ends = [self.n_params] + cut[:-1]
        self.ranges = list(zip(cut, ends))
        self.one_range = False     # the small-batch frame path: every gradient comes out of ONE launch, so the library hands
                                   # over ONE range (csrc/gfx950_layers.hip: backward_frame)
        self.cb = None
        self.draws = []

    def set_bucket_callback(self, fn):
        self.cb = fn

    def train_fwd_bwd(self, x, y, eps, grads, out=None, seed=None, offset=0, d_offset=None):
        if eps is None:
            z = self.arch['z_dim']
            eps = torch.tensor(philox_ref.normal(x.shape[0] * z, seed, offset).reshape(x.shape[0], z))
            self.draws.append(eps.numpy().copy())
        P = O.unflatten_params(self.arch, self.params.numpy())
        L, G = O.torch_loss_and_grads(self.arch, P, x.numpy(), y.numpy(), eps.numpy(), torch.float64)
        l3 = torch.tensor([float(L['G']), float(L['D_KL']), float(L['logP'])], dtype=torch.float64)
        if out is not None:
            out.copy_(l3)
        flat = torch.tensor(np.concatenate([G[n].ravel() for n in self.names]))
        grads.fill_(float('nan'))           # a range must be complete before its callback fires
        for b, (lo, hi) in enumerate([(0, self.n_params)] if self.one_range else self.ranges):
            grads[lo:hi] = flat[lo:hi]
            if self.cb is not None:
                self.cb(b, lo, hi - lo, None)
        return out if out is not None else l3

    def adam_step(self, grads, m, v, step, lr, b1, b2, eps, grad_scale):
        p, mm, vv = O.tf_adam_step(self.params.numpy(), grads.numpy() * grad_scale, m.numpy(), v.numpy(), step,
                                   lr, b1, b2, eps)
        self.params.copy_(torch.tensor(p))
        m.copy_(torch.tensor(mm))
        v.copy_(torch.tensor(vv))


def run(rank, world, F, steps, out, mode):
    from hipvae.dp import Stepper, shard_range, rank_seed
    be = OracleBackend(SMALL_ARCH, seed=10 + rank)     # different init per rank: broadcast must fix it
    st = Stepper(be, 1e-3, 0.5, 0.999, overlap=(mode != 'flat'), seed=3)
    assert st.world == world and st.rank == rank
    st.broadcast_params()
    be.one_range = mode == 'one_range'
    ncoll = []
    if world > 1:      # count the collectives a step issues
        real_ar = dist.all_reduce

        def counting(*a, **k):
            ncoll[-1] += 1
            return real_ar(*a, **k)
        dist.all_reduce = counting
    if mode == 'seeded':
        # every rank draws its OWN sampler noise: keys differ, and step t uses counter t
        assert st.seed == rank_seed(3, rank, world)
    l3 = None
    for t in range(steps):
        x, y, eps = O.make_inputs(SMALL_ARCH, F, 50 + t)
        x[F // 2:] *= 0.25                # shards with visibly different content (and losses)
        lo, hi = shard_range(F, rank, world)
        ncoll.append(0)
        if mode == 'seeded':
            l3 = st.step(torch.tensor(x[lo:hi]), torch.tensor(y[lo:hi]))
        else:
            l3 = st.step(torch.tensor(x[lo:hi]), torch.tensor(y[lo:hi]), torch.tensor(eps[lo:hi]))
    if world > 1:
        dist.all_reduce = real_ar
        # one_range / flat: ONE collective per step (gradients + losses in one buffer); bucket: the four ranges
        assert ncoll == [1 if mode in ('one_range', 'flat') else 4] * steps, ncoll
    gl = st.mean_losses(l3)
    assert torch.allclose(gl, l3)
    if mode == 'seeded':
        np.save(out + '.draw%d.npy' % rank, np.stack(be.draws))
    if rank == 0:
        np.save(out, np.concatenate([be.params.numpy(), gl.numpy()]))


def run_cb_error(rank, world, F, out):
    """An exception inside the library's bucket callback (ctypes would print and swallow it) must surface from
    Stepper.step() with the optimiser step NOT applied; a backend that reports ranges which do not tile the
    gradient buffer must be refused too (advisor finding, round 2)."""
    from hipvae.dp import Stepper, shard_range
    be = OracleBackend(SMALL_ARCH, seed=10)
    st = Stepper(be, 1e-3, 0.5, 0.999, seed=3)
    st.broadcast_params()
    x, y, eps = O.make_inputs(SMALL_ARCH, F, 50)
    lo, hi = shard_range(F, rank, world)
    args = (torch.tensor(x[lo:hi]), torch.tensor(y[lo:hi]), torch.tensor(eps[lo:hi]))
    st.step(*args)                                   # a good step first
    p1, step1 = be.params.clone(), st.step_count
    # (1) the third range's callback fails on every rank (after two all-reduces were started)
    real = st._bucket_ready_impl
    calls = []

    def failing(bucket, off, cnt, stream):
        calls.append(bucket)
        if len(calls) == 3:
            raise ValueError('injected failure in bucket %d' % bucket)
        real(bucket, off, cnt, stream)
    st._bucket_ready_impl = failing
    try:
        st.step(*args)
        raise AssertionError('step() swallowed the callback failure')
    except RuntimeError as ex:
        assert 'NOT applied' in str(ex) and isinstance(ex.__cause__, ValueError)
    assert torch.equal(be.params, p1) and st.step_count == step1
    st._bucket_ready_impl = real
    if world > 1:
        dist.barrier()
    # (2) a backend that forgets one range
    ranges = be.ranges
    be.ranges = ranges[:2] + ranges[3:]
    try:
        st.step(*args)
        raise AssertionError('a missing gradient range went unnoticed')
    except RuntimeError as ex:
        assert 'tile' in str(ex) or 'cover' in str(ex)
    assert torch.equal(be.params, p1)
    be.ranges = ranges
    st.step(*args)                                   # and the stepper still works afterwards
    assert st.step_count == step1 + 1 and not torch.equal(be.params, p1)
    if rank == 0:
        np.save(out, be.params.numpy())


def run_trainer(rank, world, F, steps, out):
    """VAETrainer.train under N ranks with a status interval of ZERO seconds on odd ranks and 'never' on even
    ones: any collective inside the status path would pair with a gradient all-reduce of another rank
    (hang or corrupted gradients).  Also exercises checkpoint save + restore-and-continue."""
    import types
    from trainer.vae import VAETrainer
    from model.vae import LossDict
    be = OracleBackend(SMALL_ARCH, seed=10)
    be.layout = {}

    class Source(object):
        t = 0

        def next_batch(self):
            x, y, _ = O.make_inputs(SMALL_ARCH, F, 50 + self.t)
            self.t += 1
            lo, hi = (rank * F // world, (rank + 1) * F // world)
            return torch.tensor(x[lo:hi]), torch.tensor(y[lo:hi])
    loss = LossDict()
    loss.machine = types.SimpleNamespace(engine=be)
    loss.source = Source()
    arch = dict(SMALL_ARCH)
    arch['training'] = dict(SMALL_ARCH['training'], max_iter=steps)
    logdir = out + '.logdir'
    args = types.SimpleNamespace(seed=0, restore_from=None, ckpt=None)
    tr = VAETrainer(loss, arch, args, {'logdir': logdir, 'logdir_root': logdir, 'restore_from': logdir})
    ck = tr.train(steps, status_secs=0 if rank % 2 else 1e9, save_secs=1e9)
    if rank == 0:
        assert os.path.basename(ck) == 'model.ckpt-%d' % steps
        msgs = open(os.path.join(logdir, 'training.log')).read().strip().splitlines()
        assert msgs and msgs[-1].startswith('Iter %05d: log P(x|z, y) = ' % steps)
    if world > 1:
        dist.barrier()
    # restore and continue: two more steps from the checkpoint == uninterrupted run
    be2 = OracleBackend(SMALL_ARCH, seed=99)
    be2.layout = {}
    loss2 = LossDict()
    loss2.machine = types.SimpleNamespace(engine=be2)
    loss2.source = Source()
    loss2.source.t = steps
    arch2 = dict(arch)
    arch2['training'] = dict(arch['training'], max_iter=steps + 2)
    args2 = types.SimpleNamespace(seed=0, restore_from=logdir, ckpt=None)
    tr2 = VAETrainer(loss2, arch2, args2, {'logdir': logdir + '2', 'logdir_root': logdir, 'restore_from': logdir})
    tr2.train(steps + 2, status_secs=1e9, save_secs=1e9)
    assert tr2.opt['g'].step_count == steps + 2
    loss.source.t = steps
    arch['training']['max_iter'] = steps + 2
    tr.train(steps + 2, status_secs=1e9, save_secs=1e9)
    assert np.abs(be.params.numpy() - be2.params.numpy()).max() < 1e-12
    if rank == 0:
        np.save(out, be.params.numpy())


def run_adv(rank, world, F, iters, out):
    """AdvStepper (hipvae/adversarial.py) under N ranks on the oracle-backed stand-ins: different initial parameters
    per rank (the broadcast must fix it), frames sharded, every gradient buffer all-reduced before its apply."""
    from adv_standin import SMALL_VAWGAN, EngineStandIn, CriticStandIn, adv_batches
    from hipvae.adversarial import AdvStepper
    from hipvae.dp import shard_range
    be, cr = EngineStandIn(SMALL_VAWGAN, 10 + rank), CriticStandIn(SMALL_VAWGAN, 20 + rank)
    st = AdvStepper(be, cr, 1e-3, 0.5, 0.999, 50.0, 10.0)
    assert st.world == world and st.rank == rank
    st.broadcast_params()
    p0 = torch.cat([be.params, cr.params]).clone()
    n_d = 2
    lo, hi = shard_range(F, rank, world)
    tt = lambda a: torch.tensor(a[lo:hi])
    it = iter(adv_batches(SMALL_VAWGAN, F, iters * (n_d + 1), 70))
    for _ in range(iters):
        for _ in range(n_d):
            b = next(it)
            st.critic_step(tt(b['x']), tt(b['y']), tt(b['eps']), tt(b['u']))
        b = next(it)
        st.generator_step(tt(b['x']), tt(b['y']), tt(b['eps']))
    if rank == 0:
        tail = np.array([float(st.status[k]) for k in ('D_KL', 'logP', 'W_dist', 'gp')])
        np.save(out, np.concatenate([(torch.cat([be.params, cr.params]) - p0).numpy(), tail]))


def run_adv_trainer(rank, world, F, iters, out):
    """VAWGANTrainer.train under N ranks on the stand-ins: nIterD + 1 batches per iteration, rank-dependent status
    intervals, status line format, checkpoint name, restore + continue == uninterrupted run (both parameter sets, the Adam
    slots and the shared apply counter travel in the checkpoint)."""
    import types
    from adv_standin import SMALL_VAWGAN, EngineStandIn, CriticStandIn
    from trainer.vae import VAWGANTrainer
    from model.vae import LossDict

    class Source(object):
        def __init__(self):
            self.t = 0

        def next_batch(self):
            x, y, _ = O.make_inputs(SMALL_VAWGAN, F, 80 + self.t)
            self.t += 1
            lo, hi = (rank * F // world, (rank + 1) * F // world)
            return torch.tensor(x[lo:hi]), torch.tensor(y[lo:hi])

    def make(seed_e, seed_c, max_iter, logdir, restore_from=None, t0=0):
        be, cr = EngineStandIn(SMALL_VAWGAN, seed_e), CriticStandIn(SMALL_VAWGAN, seed_c)
        loss = LossDict()
        loss.machine = types.SimpleNamespace(engine=be, critic=cr)
        loss.source = Source()
        loss.source.t = t0
        arch = dict(SMALL_VAWGAN)
        arch['training'] = dict(SMALL_VAWGAN['training'], max_iter=max_iter, lr=1e-3)
        args = types.SimpleNamespace(seed=0, restore_from=restore_from, ckpt=None)
        tr = VAWGANTrainer(loss, arch, args, {'logdir': logdir, 'logdir_root': logdir, 'restore_from': restore_from or logdir})
        return tr, be, cr, loss
    n_d = SMALL_VAWGAN['training']['nIterD']
    logdir = out + '.logdir'
    tr, be, cr, loss = make(10 + rank, 20 + rank, iters, logdir)     # different init per rank: the broadcast fixes it
    ck = tr.train(iters, status_secs=0 if rank % 2 else 1e9, save_secs=1e9)
    st = tr.opt['g']
    assert loss.source.t == iters * (n_d + 1) and st.step_count == iters and st.applies == iters * (n_d + 2)
    if rank == 0:
        assert os.path.basename(ck) == 'model.ckpt-%d' % iters
        msgs = open(os.path.join(logdir, 'training.log')).read().strip().splitlines()
        assert msgs and msgs[-1].startswith('Iter %05d: W_dist = ' % iters) and ' GP = ' in msgs[-1]
    if world > 1:
        dist.barrier()
    # restore into differently initialised stand-ins and continue one iteration == the uninterrupted run
    tr2, be2, cr2, loss2 = make(98, 99, iters + 1, logdir + '2', restore_from=logdir, t0=iters * (n_d + 1))
    tr2.train(iters + 1, status_secs=1e9, save_secs=1e9)
    assert tr2.opt['g'].step_count == iters + 1 and tr2.opt['g'].applies == (iters + 1) * (n_d + 2)
    tr.arch['training']['max_iter'] = iters + 1
    tr.train(iters + 1, status_secs=1e9, save_secs=1e9)
    assert np.abs(be.params.numpy() - be2.params.numpy()).max() < 1e-12
    assert np.abs(cr.params.numpy() - cr2.params.numpy()).max() < 1e-12
    if rank == 0:
        np.save(out, np.concatenate([be.params.numpy(), cr.params.numpy()]))


if __name__ == '__main__':
    torch.set_num_threads(1)
    rank, world, port, F, steps, out = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]),
                                        int(sys.argv[5]), sys.argv[6])
    mode = sys.argv[7] if len(sys.argv) > 7 else 'bucket'
    if world > 1:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = port
        os.environ['RANK'], os.environ['WORLD_SIZE'] = str(rank), str(world)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    if mode == 'trainer':
        run_trainer(rank, world, F, steps, out)
    elif mode == 'cb_error':
        run_cb_error(rank, world, F, out)
    elif mode == 'adv':
        run_adv(rank, world, F, steps, out)
    elif mode == 'adv_trainer':
        run_adv_trainer(rank, world, F, steps, out)
    else:
        run(rank, world, F, steps, out, mode)
    if world > 1:
        dist.destroy_process_group()
