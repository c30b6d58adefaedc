"""GPU tests of the runtime side of the C-ABI: per-context state and threading, the on-device sampler draw,
the fused gather + unpack of the record store, id validation, the summary reductions, the gradient-bucket
callback and a single-rank RCCL step."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

from helpers import ROOT, load_arch, rel_err
from oracle import convvae_oracle as O
from oracle import philox_ref

pytestmark = pytest.mark.gpu


def make_engine(**kw):
    from hipvae import Engine
    return Engine(load_arch(), **kw)


def test_device_sampler_draw_matches_numpy_philox():
    """vaenpvc_philox_normal and the draw inside the seeded train step are the Philox4x32-10 + Box-Muller
    sequence restated in oracle/philox_ref.py (block function pinned by the Random123 vectors)."""
    from hipvae import lib as L
    eng = make_engine()
    F, seed, off = 300, 0x123456789ABCDEF, 7
    got = eng.philox_normal(F, seed, off).cpu().numpy()
    want = philox_ref.normal(F * 128, seed, off).reshape(F, 128)
    assert np.abs(got - want).max() < 2e-5
    # moments of a larger draw
    big = eng.philox_normal(8192, 5, 0)
    assert abs(big.mean().item()) < 5e-3 and abs(big.std().item() - 1) < 5e-3
    # the seeded train step == the injected-eps train step on that draw, and it leaves the draw in the workspace
    arch = load_arch()
    P = O.init_params(arch, 2)
    x, y, _ = O.make_inputs(arch, F, 2)
    eng.load_flat(O.flatten_params(P))
    xt, yt = torch.tensor(x, device=eng.device), torch.tensor(y, device=eng.device)
    g1 = torch.zeros(eng.n_params, device=eng.device)
    g2 = torch.zeros(eng.n_params, device=eng.device)
    l1 = eng.train_fwd_bwd(xt, yt, None, g1, seed=seed, offset=off).clone()
    ws_eps = eng.ws_region(F, L.MODE_TRAIN, 'eps').view(F, 128).clone()
    assert torch.equal(ws_eps.cpu(), torch.tensor(got))
    l2 = eng.train_fwd_bwd(xt, yt, torch.tensor(got, device=eng.device), g2).clone()
    assert torch.allclose(l1, l2, rtol=1e-6)
    assert rel_err(g1.cpu().numpy(), g2.cpu().numpy()) < 1e-5
    # against the oracle on the NumPy draw
    R = O.np_forward(arch, P, x, y, want)
    assert abs(l1[0].item() - R['G']) < 1e-4 * abs(R['G'])
    # a device-side counter offsets the draw (graph replay): offset 3 + *d_off 4 == offset 7
    d_off = torch.tensor([4], dtype=torch.int64, device=eng.device)
    eng.train_fwd_bwd(xt, yt, None, g1, seed=seed, offset=3, d_offset=d_off)
    assert torch.equal(eng.ws_region(F, L.MODE_TRAIN, 'eps').view(F, 128).cpu(), torch.tensor(got))


def test_gather_unpack_records_and_id_validation():
    """analyzer.py:113-135: dequeue (gather by record number) + sp slice + Tanhize + bit-exact int64 speaker cast
    in one kernel; ids outside [0, y_dim) are reported."""
    from hipvae import HipVaeError
    eng = make_engine()
    rng = np.random.default_rng(1)
    N = 5000
    rec = rng.standard_normal((N, 1029)).astype(np.float32)
    rec[:, :513] = rng.uniform(-14, -2, (N, 513))
    spk = rng.integers(0, 10, N)
    rec[:, -1] = spk
    xmin = rng.uniform(-12, -8, 513).astype(np.float32)
    xmax = xmin + rng.uniform(2, 6, 513).astype(np.float32)
    dev = eng.device
    drec, tmin, tmax = torch.tensor(rec, device=dev), torch.tensor(xmin, device=dev), torch.tensor(xmax, device=dev)
    idx = rng.integers(0, N, 777)
    x, y = eng.unpack_records(drec, tmin, tmax, index=torch.tensor(idx, device=dev))
    xa, ya = eng.unpack_records(drec, tmin, tmax)
    assert y.dtype == torch.int64 and np.array_equal(y.cpu().numpy(), spk[idx].astype(np.int64))
    assert torch.equal(x, xa[torch.tensor(idx, device=dev)])          # gather commutes with the unpack, bit for bit
    want = O.tanhize_forward(rec[idx, :513].astype(np.float64), xmin.astype(np.float64), xmax.astype(np.float64))
    assert np.abs(x.cpu().numpy() - want).max() < 2e-6
    eng.validate_ids(y)
    bad = y.clone()
    bad[5], bad[9] = 10, -1
    with pytest.raises(HipVaeError, match='2 speaker id'):
        eng.validate_ids(bad)
    # out-of-range ids are clamped by the kernels (no fault, finite output)
    z = torch.zeros(bad.numel(), 128, device=dev)
    eng.init_params(0)
    assert torch.isfinite(eng.decode(z, bad)).all()


def test_summary_reductions_match_numpy():
    """tf.summary.histogram payload (model/vae.py:134-135): min / max / sum / sum of squares / TensorFlow's default
    bucket counts computed on the GPU."""
    from util.summary import default_bucket_limits
    eng = make_engine()
    rng = np.random.default_rng(3)
    v = np.concatenate([rng.uniform(-1, 1, 200000), rng.standard_normal(5000) * 1e-6, [0.0, 0.0, 1.0, -1.0]]).astype(np.float32)
    lim = default_bucket_limits()
    lim32 = lim.astype(np.float32)
    stats, counts = eng.summary(torch.tensor(v, device=eng.device), torch.tensor(lim32, device=eng.device))
    stats, counts = stats.cpu().numpy(), counts.cpu().numpy()
    v64 = v.astype(np.float64)
    assert stats[0] == v64.min() and stats[1] == v64.max()
    assert abs(stats[2] - v64.sum()) < 1e-6 * np.abs(v64).sum() and abs(stats[3] - (v64 ** 2).sum()) < 1e-9 * (v64 ** 2).sum()
    want = np.bincount(np.searchsorted(lim32, v, side='right'), minlength=len(lim) + 1)
    assert counts.sum() == v.size and np.array_equal(counts, want)


def test_two_contexts_on_two_host_threads():
    """No mutable state outside the context: two engines with DIFFERENT masks, precisions and timers, driven
    concurrently from two host threads on their own streams, both reproduce the oracle."""
    arch = load_arch()
    P = O.init_params(arch, 4)
    F = 96
    x, y, eps = O.make_inputs(arch, F, 4)
    L_, G = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64)
    cfgs = [dict(masks=(0xffffffff, 0xffffffff), precision='bf16x2', tag='frame_fwd'),      # (96 frames: the whole-frame kernels)
            dict(masks=(0xbfffffff, 0x000007ff), precision='bf16x3', tag='enc4_fwd')]
    engines, results, errors = [], [None, None], []
    for c in cfgs:
        e = make_engine(precision=c['precision'])
        e.set_tuned_masks(*c['masks'])
        e.load_flat(O.flatten_params(P))
        engines.append(e)

    def work(i):
        try:
            e = engines[i]
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                xt, yt, et = (torch.tensor(x, device=e.device), torch.tensor(y, device=e.device), torch.tensor(eps, device=e.device))
                g = torch.zeros(e.n_params, device=e.device)
                e.timer_select(cfgs[i]['tag'])
                for _ in range(20):
                    l3 = e.train_fwd_bwd(xt, yt, et, g).clone()
                s.synchronize()
                results[i] = (l3.cpu().numpy(), g.cpu().numpy(), e.timer_read())
        except Exception as ex:      # noqa: BLE001
            errors.append(ex)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for i, (l3, g, (ms, n)) in enumerate(results):
        assert n == 20 and ms > 0                      # each context counted only its own tagged kernel
        assert abs(l3[0] - L_['G']) < 1e-4 * abs(L_['G'])
        for name, (off, shape) in engines[i].layout.items():
            k = int(np.prod(shape))
            assert rel_err(g[off:off + k].reshape(shape), G[name]) < 2e-4, (i, name)
    assert engines[0].precision == 2 and engines[1].precision == 3


def test_backward_only_call_needs_a_matching_train_step():
    """vaenpvc_train_bwd_target consumes the activations, packed weights and operand planes the preceding train step of the
    same context left in the same workspace: without one -- or after one of another batch size or kernel selection -- it
    must fail with VAENPVC_E_STATE instead of reading stale operands (round-3 advisor finding)."""
    from hipvae import lib as L
    arch = load_arch()
    eng = make_engine()
    eng.init_params(2)
    x, y, eps = O.make_inputs(arch, 24, 2)
    xt, yt, et = (torch.tensor(a, device=eng.device) for a in (x, y, eps))
    grads = torch.zeros(eng.n_params, device=eng.device)
    with pytest.raises(L.HipVaeError, match='no matching train step'):
        eng.train_bwd_target(xt, yt, et, xt, grads)                    # nothing precedes it
    eng.train_fwd_bwd(xt, yt, et, grads)
    g1 = grads.clone()
    eng.train_bwd_target(xt, yt, et, xt, grads)                        # same batch, target = x: the same gradient
    torch.cuda.synchronize()
    assert ((grads - g1).abs().max() / g1.abs().max()).item() < 1e-5
    with pytest.raises(L.HipVaeError, match='no matching train step'):
        eng.train_bwd_target(xt[:16], yt[:16], et[:16], xt[:16], grads)   # another batch size
    # a forward-only entry point on the same context may overwrite activations / planes of the train step (round-4 advisor)
    for fwd_only in (lambda: eng.loss_fwd(xt, yt, et), lambda: eng.encode(xt)):
        eng.train_fwd_bwd(xt, yt, et, grads)
        fwd_only()
        with pytest.raises(L.HipVaeError, match='no matching train step'):
            eng.train_bwd_target(xt, yt, et, xt, grads)
    eng.train_fwd_bwd(xt, yt, et, grads)
    eng.train_bwd_target(xt, yt, et, xt, grads)                        # ... and a fresh train step re-arms it
    eng.set_tuned_masks(0xffffffff & ~(1 << 21), 0xffffffff & ~(1 << 21))
    with pytest.raises(L.HipVaeError, match='no matching train step'):
        eng.train_bwd_target(xt, yt, et, xt, grads)                    # another kernel family


@pytest.mark.parametrize('path', ['layered', 'frame'])
def test_gradient_bucket_callback_ranges_and_order(path):
    """vaenpvc_set_bucket_callback.  Layered kernels (every batch above 512 frames; forced here by clearing mask bit 21): four
    contiguous ranges, reported back to front as the backward pass finishes them, tiling the flat buffer exactly.  Small-batch
    frame kernels (the default at this batch size): every gradient comes out of one launch, so ONE range covers the buffer
    (one all-reduce per step, not four back-to-back ones).  Either way a range is complete on the stream handed to the
    callback (checked by copying it there and then)."""
    arch = load_arch()
    eng = make_engine()
    if path == 'layered':
        eng.set_tuned_masks(0xffffffff & ~(1 << 21), 0xffffffff & ~(1 << 21))
    eng.init_params(1)
    F = 64
    x, y, eps = O.make_inputs(arch, F, 1)
    xt, yt, et = torch.tensor(x, device=eng.device), torch.tensor(y, device=eng.device), torch.tensor(eps, device=eng.device)
    grads = torch.zeros(eng.n_params, device=eng.device)
    seen, snaps = [], []

    def cb(bucket, off, cnt, stream):
        seen.append((bucket, off, cnt))
        ext = torch.cuda.ExternalStream(int(stream)) if stream else torch.cuda.current_stream()
        with torch.cuda.stream(ext):
            snaps.append(grads[off:off + cnt].clone())
    eng.set_bucket_callback(cb)
    eng.train_fwd_bwd(xt, yt, et, grads)
    torch.cuda.synchronize()
    eng.set_bucket_callback(None)
    offs = {n: o for n, (o, _) in eng.layout.items()}
    if path == 'layered':
        want = [(offs['Generator/conv2d_transpose/kernel'], eng.n_params),
                (offs['Generator/fully_connected/weights'], offs['Generator/conv2d_transpose/kernel']),
                (offs['Encoder/dense/kernel'], offs['Generator/fully_connected/weights']),
                (0, offs['Encoder/dense/kernel'])]
    else:
        want = [(0, eng.n_params)]
    assert [b for b, _, _ in seen] == list(range(len(want)))
    assert [(o, o + c) for _, o, c in seen] == want
    for (_, off, cnt), snap in zip(seen, snaps):
        assert torch.equal(snap, grads[off:off + cnt])
    assert grads.abs().max().item() > 0
    n0 = len(seen)
    eng.train_fwd_bwd(xt, yt, et, grads)
    assert len(seen) == n0                             # unregistered


RCCL_STEP = r"""
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, 'vae-npvc_amd')); sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29641', RANK='0', WORLD_SIZE='1', VAENPVC_FORCE_DIST='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from hipvae import Engine
from hipvae.dp import Stepper
from oracle import convvae_oracle as O
arch = json.load(open(os.path.join(%(root)r, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
P = O.init_params(arch, 3); x, y, eps = O.make_inputs(arch, 64, 3)
out = {}
for mode in ('plain', 'bucket', 'flat', 'graph'):
    os.environ['VAENPVC_FORCE_DIST'] = '0' if mode == 'plain' else '1'
    eng = Engine(arch); eng.load_flat(O.flatten_params(P))
    st = Stepper(eng, 1e-4, 0.5, 0.999, overlap=(mode != 'flat'))
    assert st.collective == (mode != 'plain')
    xt, yt, et = (torch.tensor(a, device=eng.device) for a in (x, y, eps))
    if mode == 'graph':
        st.capture(xt, yt, et)
    for t in range(3):
        l3 = st.replay() if mode == 'graph' else st.step(xt, yt, et)
        if t == 0:
            g1, p1 = st.grads.clone(), eng.params.clone()
    l3 = st.mean_losses() if mode == 'graph' else l3
    torch.cuda.synchronize()
    out[mode] = (eng.params.cpu().numpy(), l3.cpu().numpy(), g1.cpu().numpy(), p1.cpu().numpy())
p0 = O.flatten_params(P)
ref = out['plain']
strong = np.abs(ref[2]) > 1e-2 * np.abs(ref[2]).max()
for mode in ('bucket', 'flat', 'graph'):
    r = out[mode]
    # first step: same parameters, so only the order of fp32 atomics differs; later steps: see test_hipgraph_replay_matches_eager
    assert np.abs(r[2] - ref[2]).max() <= 1e-5 * np.abs(ref[2]).max(), mode
    assert np.abs((r[3] - p0) - (ref[3] - p0))[strong].max() <= 1e-3 * np.abs(ref[3] - p0).max(), mode
    assert np.abs((r[0] - p0) - (ref[0] - p0))[strong].max() <= 5e-2 * np.abs(ref[0] - p0).max(), mode
    assert np.abs(r[0] - ref[0]).mean() <= 5e-3 * np.abs(ref[0] - p0).max(), mode
    assert np.allclose(r[1], ref[1], rtol=1e-4), mode
dist.destroy_process_group()
print('RCCL_OK')
"""


def test_single_rank_rccl_step_bucketed_flat_and_captured():
    """The real Engine + "nccl" (RCCL) path of hipvae.dp with one rank (VAENPVC_FORCE_DIST=1): bucketed all-reduce
    from the library's callback, one flat all-reduce, and the all-reduce captured in a hipGraph all follow the
    collective-free trajectory."""
    r = subprocess.run([sys.executable, '-c', RCCL_STEP % {'root': ROOT}], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
