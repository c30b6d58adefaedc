"""Golden fixtures at the BENCHMARKED batch sizes (F = 8 192 and F = 32 768 frames) from the
float64 CPU oracle, evaluated in chunks of 256 frames.

    python tests/golden/make_golden_large.py            # ~5 min on 8 cores

The loss is a batch mean over independent frames (model/vae.py:112-128; per-sample LayerNorm,
util/layers.py:32), so with equal chunks
    loss(F frames)      = mean over chunks of loss(chunk)
    d loss / d params   = mean over chunks of d loss(chunk) / d params
which lets the float64 autograd oracle cover batches it could not hold at once.

PARITY UNPINNED (see make_golden.py): outputs of OUR restatement.  The fixture stores outputs
only -- per-tensor gradient L2 norms / abs-max / 64 sampled entries, the three losses, and the
z_mu / z_lv / xh rows of 16 sampled frames; inputs and weights are regenerated from the seed.

KINK-SAFE INPUTS.  lrelu (util/layers.py:147-149) is not differentiable at 0: where a LayerNorm output n
lies within rounding of 0, a float32 evaluation (ours, or TensorFlow's) and the float64 oracle can land on
different sides and take slopes 1 and 0.02 -- a finite difference in that frame's gradient that no amount
of arithmetic care removes.  One train step evaluates 18 540 such units per frame: at 256 frames a flip is
rare (measured gradient errors ~1e-6), at 8 192+ frames several happen in every batch and put a floor of
~2e-4 under ANY fp32 implementation's error against float64 (measured: hand-written fp32 kernels 1.8e-4,
geometry-generic kernels 1.5e-4, PyTorch-CPU fp32 8e-4 on the unfiltered seed-21 batch, each with its own
flips, identical whether the batch is evaluated at once or in chunks).  A parity test must not depend on
which side of a kink a rounding error falls, so the fixture's batch is built from a seeded candidate stream of
3F frames, keeping the first F whose float64 forward pass has every |n| >= TAU (5e-5: about half of the frames
qualify); the kept candidate numbers are stored (`frame_src`) and the test rebuilds the batch from the same
stream.  (At the small batch sizes, where the oracle runs inside the test, the comparison instead pins the branch
of every near-kink unit to the one the GPU took: oracle.torch_lrelu.)

It also records how far FLOAT32 arithmetic itself is from these float64 values: the same chunks are
evaluated with the oracle's PyTorch-CPU restatement in float32 (the stand-in for the reference's
fp32 TensorFlow path) and the per-tensor error of ITS gradient on the same sampled entries is stored
as `ref32_grad_err` (relative to the tensor's abs-max, like the test's metric).  At these batch sizes
that error is 1e-4 .. 1e-3 on the first encoder layers (rounding amplified through ~20 layers with
LayerNorm), i.e. above the 2e-4 gradient bar: no fp32 implementation can be held to the bar there, and the
GPU test then asks for at most HALF the stand-in's error instead.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from helpers import load_arch, sample_idx  # noqa: E402
from oracle import convvae_oracle as O  # noqa: E402

CHUNK = 256
N_GRAD_SAMPLES = 64
N_FRAME_SAMPLES = 16
TAU = 5e-5      # kink margin on the LayerNorm outputs (fp32 evaluations of them differ by ~1e-6, 2-term split ones by ~1e-5)
CAND = 3        # candidate stream = CAND * F frames (about half of the frames have a unit inside the margin)


def min_abs_preactivation(arch, P64, x, y, eps):
    """per frame: min |n| over all LayerNorm outputs n (the lrelu inputs) of the float64 forward pass"""
    xt, yt, et = torch.tensor(x, dtype=torch.float64), torch.tensor(y), torch.tensor(eps, dtype=torch.float64)
    z_mu, z_lv, acts = O.torch_encode(arch, P64, xt)
    out = torch.full((x.shape[0],), float('inf'), dtype=torch.float64)
    for i, (a, _) in enumerate(acts):
        p = 'Encoder/Conv2d-%d/' % i
        n = O.torch_layernorm(a, P64[p + 'layernorm.offset'], P64[p + 'layernorm.scale'])
        out = torch.minimum(out, n.abs().reshape(x.shape[0], -1).min(dim=1).values)
    z = z_mu + et * torch.sqrt(torch.exp(z_lv))
    _, dacts = O.torch_decode(arch, P64, z, yt)
    for i, a in enumerate(dacts[1:]):
        n = O.torch_layernorm(a, P64['Generator/ConvT-LN%d.offset' % i], P64['Generator/ConvT-LN%d.scale' % i])
        out = torch.minimum(out, n.abs().reshape(x.shape[0], -1).min(dim=1).values)
    return out.numpy()


def kink_safe_inputs(arch, P, F, seed):
    """(x, y, eps, frame_src): the first F frames of the seeded 3F/2-candidate stream whose lrelu inputs all keep
    at least TAU away from 0."""
    Fc = F * CAND
    xc, yc, ec = O.make_inputs(arch, Fc, seed)
    P64 = O.torch_params(P, torch.float64)
    keep = []
    with torch.no_grad():
        for c in range(0, Fc, CHUNK):
            sl = slice(c, min(Fc, c + CHUNK))
            m = min_abs_preactivation(arch, P64, xc[sl], yc[sl], ec[sl])
            keep.extend((c + np.nonzero(m >= TAU)[0]).tolist())
            if len(keep) >= F:
                break
    assert len(keep) >= F, 'candidate stream too short: %d of %d' % (len(keep), F)
    idx = np.array(keep[:F], np.int32)
    print('F=%d: kept %d of the first %d candidates' % (F, F, idx[-1] + 1), flush=True)
    return xc[idx], yc[idx], ec[idx], idx, Fc


def run(arch, F, seed):
    assert F % CHUNK == 0
    P = O.init_params(arch, seed)
    x, y, eps, frame_src, n_cand = kink_safe_inputs(arch, P, F, seed)
    names = list(P.keys())
    gsum = {n: np.zeros(P[n].shape, np.float64) for n in names}
    g32sum = {n: np.zeros(P[n].shape, np.float64) for n in names}
    lsum = np.zeros(3, np.float64)
    fidx = sample_idx(F, N_FRAME_SAMPLES)
    rows = {'z_mu': {}, 'z_lv': {}, 'xh': {}}
    t0 = time.time()
    for c in range(F // CHUNK):
        sl = slice(c * CHUNK, (c + 1) * CHUNK)
        L, G = O.torch_loss_and_grads(arch, P, x[sl], y[sl], eps[sl], torch.float64)
        _, G32 = O.torch_loss_and_grads(arch, P, x[sl], y[sl], eps[sl], torch.float32)
        for n in names:
            gsum[n] += G[n]
            g32sum[n] += G32[n].astype(np.float64)
        lsum += np.array([L['G'], L['D_KL'], L['logP']], np.float64)
        for f in fidx:
            if sl.start <= f < sl.stop:
                for k in rows:
                    rows[k][int(f)] = L[k][f - sl.start].copy()
        if c % 16 == 0:
            print('F=%d chunk %d/%d  %.0fs' % (F, c, F // CHUNK, time.time() - t0), flush=True)
    nch = F // CHUNK
    out = {'loss3': lsum / nch, 'frame_idx': fidx, 'frame_src': frame_src, 'n_candidates': np.int64(n_cand),
           'tau': np.float64(TAU)}
    for k in rows:
        out[k + '_rows'] = np.stack([rows[k][int(f)] for f in fidx])
    G = {n: gsum[n] / nch for n in names}
    out['grad_l2'] = np.array([np.sqrt((G[n] ** 2).sum()) for n in names])
    out['grad_absmax'] = np.array([np.abs(G[n]).max() for n in names])
    out['grad_samples'] = np.stack([
        np.pad(G[n].ravel()[sample_idx(G[n].size, N_GRAD_SAMPLES)], (0, N_GRAD_SAMPLES - min(N_GRAD_SAMPLES, G[n].size)))
        for n in names])
    out['ref32_grad_err'] = np.array([
        np.abs((g32sum[n] / nch).ravel()[sample_idx(G[n].size, N_GRAD_SAMPLES)] - G[n].ravel()[sample_idx(G[n].size, N_GRAD_SAMPLES)]).max()
        / max(np.abs(G[n]).max(), 1e-300) for n in names])
    return out


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count() or 1)
    arch = load_arch()
    cases = [(8192, 21), (32768, 22)]
    if len(sys.argv) > 1:
        cases = [(int(a.split(':')[0]), int(a.split(':')[1])) for a in sys.argv[1:]]
    for F, seed in cases:
        np.savez_compressed(os.path.join(HERE, 'vcc2016_F%d_seed%d.npz' % (F, seed)), **run(arch, F, seed))
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))
