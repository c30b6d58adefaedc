"""Developer diagnostic (GPU box): where does the gradient error at a large batch come from?  Runs the golden
F = 8192 case under several kernel selections / precisions and prints the worst gradient errors."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vae-npvc_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from helpers import load_arch, sample_idx, GOLDEN, golden_large_inputs  # noqa: E402
from oracle import convvae_oracle as O  # noqa: E402
from hipvae import Engine  # noqa: E402

F, seed = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 21)
arch = load_arch()
gold = np.load(os.path.join(GOLDEN, 'vcc2016_F%d_seed%d.npz' % (F, seed)))
P = O.init_params(arch, seed)
x, y, eps = golden_large_inputs(arch, gold, F, seed)
CFG = [('generic kernels', dict(precision='bf16x3', impl='generic'), (0xffffffff, 0xffffffff), {}),
       ('fp32, no side stream', dict(precision='bf16x3'), (0x9fffffff | (1 << 30), 0x9fffffff), {'VAENPVC_TOEP': 'f32'}),
       ('fp32 kernels only', dict(precision='bf16x3'), (0x9fffffff | (1 << 30), 0x9fffffff | (1 << 30)), {'VAENPVC_TOEP': 'f32'}),
       ('toep x3, dense fp32', dict(precision='bf16x3'), (0xdfffffff, 0xdfffffff), {}),
       ('toep x2, dense fp32', dict(precision='auto'), (0xdfffffff, 0xdfffffff), {}),
       ('toep x3, dense x3', dict(precision='bf16x3'), (0xffffffff, 0xffffffff), {}),
       ('auto', dict(precision='auto'), (0xffffffff, 0xffffffff), {}),
       ('auto, no side stream', dict(precision='auto'), (0xffffffff, 0xbfffffff), {})]
for name, kw, masks, env in CFG:
    os.environ.pop('VAENPVC_TOEP', None)
    os.environ.update(env)
    eng = Engine(arch, **kw)
    eng.set_tuned_masks(*masks)
    eng.load_flat(O.flatten_params(P))
    xt, yt, et = (torch.tensor(a, device=eng.device) for a in (x, y, eps))
    g = torch.zeros(eng.n_params, device=eng.device)
    eng.train_fwd_bwd(xt, yt, et, g)
    g = g.cpu().numpy().astype(np.float64)
    errs = []
    for i, (n, (off, shape)) in enumerate(eng.layout.items()):
        k = int(np.prod(shape))
        e = np.abs(g[off:off + k][sample_idx(k, 64)] - gold['grad_samples'][i][:min(64, k)]).max() / max(gold['grad_absmax'][i], 1e-12)
        errs.append((e, n))
    errs.sort(reverse=True)
    print('absmax enc0 kernel %.3e  l2 %.3e' % (gold['grad_absmax'][1], gold['grad_l2'][1])) if name.startswith('generic') else None
    print('%-24s' % name, '  '.join('%.2e %s' % (e, n.split('/')[-2] + '/' + n.split('/')[-1]) for e, n in errs[:4]), flush=True)
    del eng
