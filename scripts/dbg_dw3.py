"""Developer tool: where the 1025-tap layer's weight gradient of the tuned path differs from the generic path (GPU vs GPU).
usage: python scripts/dbg_dw3.py F"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import convvae_oracle as O
from hipvae import Engine
arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
F = int(sys.argv[1])
P = O.init_params(arch, 5)
x, y, eps = O.make_inputs(arch, F, 3)
res = {}
for impl in ('generic', 'auto'):
    eng = Engine(arch, impl=impl)
    eng.load_flat(O.flatten_params(P))
    dev = eng.device
    g = torch.full((eng.n_params,), float('nan'), device=dev)
    eng.train_fwd_bwd(torch.tensor(x, device=dev), torch.tensor(y, device=dev), torch.tensor(eps, device=dev), g)
    torch.cuda.synchronize()
    off, shape = eng.layout['Generator/conv2d_transpose_3/kernel']
    res[impl] = g[off:off + 1025 * 8].cpu().numpy().reshape(1025, 8)
d = np.abs(res['auto'] - res['generic'])
den = np.abs(res['generic']).max()
print('F', F, 'worst', d.max() / den, 'at (tap, channel)', np.unravel_index(d.argmax(), d.shape))
rows = d.max(axis=1) / den
bad = np.flatnonzero(rows > 2e-5)
print('taps over 2e-5:', len(bad), bad[:20], '...', bad[-10:] if len(bad) else '')
print('per-channel worst', d.max(axis=0) / den)
