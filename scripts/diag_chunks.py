"""Developer diagnostic (GPU box): is the large-batch gradient error made by the frame reductions?  The F = 8192
golden case is evaluated (a) in one call and (b) as 32 calls of 256 frames averaged in float64 on the host."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vae-npvc_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from helpers import load_arch, sample_idx, GOLDEN  # noqa: E402
from oracle import convvae_oracle as O  # noqa: E402
from hipvae import Engine  # noqa: E402
F, seed = 8192, 21
arch = load_arch()
gold = np.load(os.path.join(GOLDEN, 'vcc2016_F%d_seed%d.npz' % (F, seed)))
P = O.init_params(arch, seed)
x, y, eps = O.make_inputs(arch, F, seed)
os.environ['VAENPVC_TOEP'] = 'f32'
eng = Engine(arch, precision='bf16x3')
eng.set_tuned_masks(0xdfffffff, 0xdfffffff)
eng.load_flat(O.flatten_params(P))
xt, yt, et = (torch.tensor(a, device=eng.device) for a in (x, y, eps))


def errs(g, tag):
    out = []
    for i, (n, (off, shape)) in enumerate(eng.layout.items()):
        k = int(np.prod(shape))
        e = np.abs(g[off:off + k][sample_idx(k, 64)] - gold['grad_samples'][i][:min(64, k)]).max() / max(gold['grad_absmax'][i], 1e-12)
        out.append((e, n))
    out.sort(reverse=True)
    print('%-28s' % tag, '  '.join('%.2e %s' % (e, n.split('/')[-2] + '/' + n.split('/')[-1]) for e, n in out[:5]), flush=True)


g = torch.zeros(eng.n_params, device=eng.device)
eng.train_fwd_bwd(xt, yt, et, g)
errs(g.cpu().numpy().astype(np.float64), 'one call, F = 8192')
for C in (256, 1024):
    acc = np.zeros(eng.n_params, np.float64)
    for c in range(F // C):
        sl = slice(c * C, (c + 1) * C)
        eng.train_fwd_bwd(xt[sl], yt[sl], et[sl], g)
        acc += g.cpu().numpy().astype(np.float64)
    errs(acc / (F // C), '%d calls of %d, f64 mean' % (F // C, C))
