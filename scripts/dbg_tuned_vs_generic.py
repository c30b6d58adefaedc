"""Developer tool: per-tensor gradient difference between the tuned and the generic path
(GPU vs GPU, same inputs) for a given F and a structured (large mean / small variance) input."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import convvae_oracle as O
from hipvae import Engine
import json
arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mode = sys.argv[2] if len(sys.argv) > 2 else 'struct'
P = O.init_params(arch, 5)
x, y, eps = O.make_inputs(arch, F, 3)
if mode == 'struct':
    rng = np.random.default_rng(0)
    x = (0.8 + 0.01 * rng.standard_normal(x.shape)).astype(np.float32).clip(-1, 1)
    y[:] = np.where(np.arange(F) % 2 == 0, 0, 9)
res = {}
for impl in ('generic', 'auto'):
    eng = Engine(arch, impl=impl)
    eng.load_flat(O.flatten_params(P))
    dev = eng.device
    g = torch.full((eng.n_params,), float('nan'), device=dev)
    l3 = eng.train_fwd_bwd(torch.tensor(x, device=dev), torch.tensor(y, device=dev), torch.tensor(eps, device=dev), g).clone()
    torch.cuda.synchronize()
    res[impl] = (l3.cpu().numpy(), g.cpu().numpy())
print('loss generic', res['generic'][0], 'tuned', res['auto'][0])
lay = O.param_layout(arch)
off = 0
for name, shp in lay.items():
    n = int(np.prod(shp))
    a = res['generic'][1][off:off + n]; b = res['auto'][1][off:off + n]
    den = np.abs(a).max() + 1e-30
    print('%-40s %-18s max|g| %.3e  rel diff %.3e' % (name, str(shp), den, np.abs(a - b).max() / den))
    off += n
