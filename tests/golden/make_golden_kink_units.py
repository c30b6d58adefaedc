"""The lrelu units of the UNFILTERED benchmarked batch (F = 32 768 frames, seed 23 -- the batch of
make_golden_unfiltered.py) whose float64 LayerNorm output n lies within KINK_TAU = 1e-4 of the kink, per layer:

    python tests/golden/make_golden_kink_units.py          # ~3 min on 8 cores (float64 forward only, chunked)

A unit can only land on the other side of lrelu's kink than float64 when |n| is below the evaluation's own error in n;
test_kink_flips_explain_the_unfiltered_gradient_excess asserts at 2 048 frames (oracle inside the test) that every
flipped unit lies within 1e-4 of the kink.  The 607 M units of this batch do not fit a fixture, the ~50 k near-kink
ones do: flat index into the layer's [F][C][H] tensor + the float64 value of n.  The GPU test recomputes n from the
GPU's own pre-LN tensors and statistics at these units and counts the branches that differ
(tests/test_gpu_parity.py::test_unfiltered_benchmark_batch_statistics), which puts a NUMBER on the kink-flip rate at the
benchmarked batch size instead of a bound inferred from the gradient error.

PARITY UNPINNED (see make_golden.py): outputs of OUR restatement.  Outputs only; inputs and weights come from the seed.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from helpers import load_arch  # noqa: E402
from oracle import convvae_oracle as O  # noqa: E402

CHUNK = 512
KINK_TAU = 1e-4
LAYERS = [('enc', i, 'Encoder/Conv2d-%d/layernorm' % i) for i in range(5)] + [('dec', i, 'Generator/ConvT-LN%d' % i) for i in range(3)]


def run(arch, F, seed):
    assert F % CHUNK == 0
    P = O.init_params(arch, seed)
    x, y, eps = O.make_inputs(arch, F, seed)
    Pt = O.torch_params(P, torch.float64)
    idx = {(net, i): [] for net, i, _ in LAYERS}
    val = {(net, i): [] for net, i, _ in LAYERS}
    units = {(net, i): 0 for net, i, _ in LAYERS}
    t0 = time.time()
    with torch.no_grad():
        for c in range(F // CHUNK):
            sl = slice(c * CHUNK, (c + 1) * CHUNK)
            xt, et = torch.tensor(x[sl], dtype=torch.float64), torch.tensor(eps[sl], dtype=torch.float64)
            z_mu, z_lv, eacts = O.torch_encode(arch, Pt, xt)
            _, dacts = O.torch_decode(arch, Pt, z_mu + et * torch.sqrt(torch.exp(z_lv)), torch.tensor(y[sl]))
            pre = {('enc', i): a for i, (a, _) in enumerate(eacts)}
            pre.update({('dec', i): a for i, a in enumerate(dacts[1:])})
            for net, i, name in LAYERS:
                a = pre[(net, i)]
                n = O.torch_layernorm(a, Pt[name + '.offset'], Pt[name + '.scale'])[..., 0].numpy()
                per = n[0].size
                near = np.flatnonzero(np.abs(n.ravel()) < KINK_TAU)
                idx[(net, i)].append(near.astype(np.int64) + c * CHUNK * per)
                val[(net, i)].append(n.ravel()[near])
                units[(net, i)] += n.size
            if c % 8 == 0:
                print('chunk %d/%d  %.0fs' % (c, F // CHUNK, time.time() - t0), flush=True)
    out = {'tau': np.float64(KINK_TAU), 'F': np.int64(F), 'seed': np.int64(seed)}
    for net, i, _ in LAYERS:
        k = '%s%d' % (net, i)
        out[k + '_idx'] = np.concatenate(idx[(net, i)]).astype(np.int32)
        out[k + '_n'] = np.concatenate(val[(net, i)])
        out[k + '_units'] = np.int64(units[(net, i)])
        print('%s: %d of %d units within %.0e of the kink' % (k, out[k + '_idx'].size, units[(net, i)], KINK_TAU))
    return out


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count() or 1)
    arch = load_arch()
    F, seed = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32768, 23)
    r = run(arch, F, seed)
    path = os.path.join(HERE, 'vcc2016_F%d_seed%d_kink_units.npz' % (F, seed))
    np.savez_compressed(path, **r)
    print(path, os.path.getsize(path))
