"""Developer diagnostic (GPU box): per-chunk gradient error against the float64 oracle for the golden-large seeds."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vae-npvc_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from helpers import load_arch  # noqa: E402
from oracle import convvae_oracle as O  # noqa: E402
from hipvae import Engine  # noqa: E402
torch.set_num_threads(16)
arch = load_arch()
os.environ['VAENPVC_TOEP'] = 'f32'
for pseed, dseed, F, C in ((21, 21, 8192, 256), (2, 2, 256, 256), (21, 2, 256, 256), (2, 21, 8192, 256), (22, 22, 32768, 256)):
    P = O.init_params(arch, pseed)
    x, y, eps = O.make_inputs(arch, F, dseed)
    x, y, eps = x[:C], y[:C], eps[:C]
    eng = Engine(arch, precision='bf16x3')
    eng.set_tuned_masks(0xdfffffff, 0xdfffffff)
    eng.load_flat(O.flatten_params(P))
    xt, yt, et = (torch.tensor(a, device=eng.device) for a in (x, y, eps))
    g = torch.zeros(eng.n_params, device=eng.device)
    eng.train_fwd_bwd(xt, yt, et, g)
    g = g.cpu().numpy().astype(np.float64)
    L, G = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64)
    _, G32 = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float32)
    errs, e32 = [], []
    for n, (off, shape) in eng.layout.items():
        k = int(np.prod(shape))
        errs.append((np.abs(g[off:off + k].reshape(shape) - G[n]).max() / np.abs(G[n]).max(), n))
        e32.append((np.abs(G32[n] - G[n]).max() / np.abs(G[n]).max(), n))
    errs.sort(reverse=True); e32.sort(reverse=True)
    print('params %d data %d (first %d of %d): GPU ' % (pseed, dseed, C, F), '  '.join('%.2e %s' % (e, n.split('/')[-2] + '/' + n.split('/')[-1]) for e, n in errs[:3]))
    print('      fp32 CPU ', '  '.join('%.2e %s' % (e, n.split('/')[-2] + '/' + n.split('/')[-1]) for e, n in e32[:3]), flush=True)
