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


def run(arch, F, seed):
    assert F % CHUNK == 0
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    names = list(P.keys())
    gsum = {n: np.zeros(P[n].shape, np.float64) for n in names}
    lsum = np.zeros(3, np.float64)
    fidx = sample_idx(F, N_FRAME_SAMPLES)
    rows = {'z_mu': {}, 'z_lv': {}, 'xh': {}}
    t0 = time.time()
    for c in range(F // CHUNK):
        sl = slice(c * CHUNK, (c + 1) * CHUNK)
        L, G = O.torch_loss_and_grads(arch, P, x[sl], y[sl], eps[sl], torch.float64)
        for n in names:
            gsum[n] += G[n]
        lsum += np.array([L['G'], L['D_KL'], L['logP']], np.float64)
        for f in fidx:
            if sl.start <= f < sl.stop:
                for k in rows:
                    rows[k][int(f)] = L[k][f - sl.start].copy()
        if c % 16 == 0:
            print('F=%d chunk %d/%d  %.0fs' % (F, c, F // CHUNK, time.time() - t0), flush=True)
    nch = F // CHUNK
    out = {'loss3': lsum / nch, 'frame_idx': fidx}
    for k in rows:
        out[k + '_rows'] = np.stack([rows[k][int(f)] for f in fidx])
    G = {n: gsum[n] / nch for n in names}
    out['grad_l2'] = np.array([np.sqrt((G[n] ** 2).sum()) for n in names])
    out['grad_absmax'] = np.array([np.abs(G[n]).max() for n in names])
    out['grad_samples'] = np.stack([
        np.pad(G[n].ravel()[sample_idx(G[n].size, N_GRAD_SAMPLES)], (0, N_GRAD_SAMPLES - min(N_GRAD_SAMPLES, G[n].size)))
        for n in names])
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
