"""Diagnostic (run by hand on the GPU box: python tests/diag_trajectory.py [steps] [F]): per-step error of an Adam run on one
fixed batch against float64, for several kernel selections.  At every step the float64 oracle is evaluated AT THE GPU's OWN
PARAMETERS, so the numbers are per-step errors (non-chaotic), not trajectory drift:
  g_raw   gradient max-norm error over the largest entry, no kink pinning
  g_pin   the same with the lrelu units within 1e-4 of the kink pinned to the GPU's branch
  flips   units whose branch differs between the GPU and float64 (all layers)
  loss    loss triple error
and the drift of the trajectory itself (loss vs the float64 run from the same start).
Also: repeatability of ONE evaluation (30 calls on identical parameters): a race shows here, atomics ordering is ~1e-6."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vae-npvc_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
from oracle import convvae_oracle as O
from test_gpu_parity import ARCHS, KINK_TAU, gpu_branches
from test_gpu_frame import oracle_adam_trajectory
from hipvae import Engine
from hipvae.dp import Stepper

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
arch = ARCHS['vcc']
FRAME = 1 << 21
CONFIGS = [('generic', 0xffffffff & ~FRAME, None, 'generic'), ('layered bf16x2', 0xffffffff & ~FRAME, None, 'auto'),
           ('frame', 0xffffffff, None, 'auto')]
P0, P1, (x, y, eps), want = oracle_adam_trajectory(arch, F, 3, N)
TR = [oracle_adam_trajectory(arch, F, 3, t)[1] for t in range(1, N + 1)] if N <= 24 else None   # f64 parameters after t steps
names = list(P0.keys())
for tag, mask, prec, impl in CONFIGS:
    eng = Engine(arch, impl=impl, precision=prec)
    eng.set_tuned_masks(mask, mask)
    eng.load_flat(O.flatten_params(P0))
    xt, yt, et = (torch.tensor(a, device=eng.device) for a in (x, y, eps))
    st = Stepper(eng, 1e-4, 0.5, 0.999)
    # repeatability on identical parameters
    gs = []
    for _ in range(30):
        eng.train_fwd_bwd(xt, yt, et, st.grads)
        torch.cuda.synchronize()
        gs.append(st.grads.clone())
    gs = torch.stack(gs)
    rep = ((gs - gs[0]).abs().max() / gs[0].abs().max()).item()
    print('%-16s repeatability of one evaluation over 30 calls: %.2e' % (tag, rep), flush=True)
    for t in range(N):
        flat = eng.params.cpu().numpy()
        P = O.unflatten_params(arch, flat)
        l3 = st.step(xt, yt, et).clone().cpu().numpy()
        g = st.grads.cpu().numpy()
        br = gpu_branches(eng, arch, P, F)
        L, G = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64)
        Lp, Gp = O.torch_loss_and_grads(arch, P, x, y, eps, torch.float64, kink=br)
        ref = np.concatenate([G[n].ravel() for n in names]); refp = np.concatenate([Gp[n].ravel() for n in names])
        # branch flips: float64 forward at these parameters
        R = O.np_forward(arch, P, x, y, eps)
        flips = 0
        for net, nl, pre in (('enc', 5, 'Encoder/Conv2d-%d/layernorm'), ('dec', 3, 'Generator/ConvT-LN%d')):
            for i in range(nl):
                a = R['%s_a%d' % (net, i)]
                mu = a.mean(axis=(1, 2), keepdims=True); rs = 1 / np.sqrt(a.var(axis=(1, 2), keepdims=True) + 1e-5)
                n = (a - mu) * rs * np.asarray(P[(pre % i) + '.scale'], np.float64).reshape(1, -1, 1) + \
                    np.asarray(P[(pre % i) + '.offset'], np.float64).reshape(1, -1, 1)
                flips += int(((n >= 0) != br['%s%d' % (net, i)]).sum())
        e_raw = np.abs(g - ref).max() / np.abs(ref).max()
        e_pin = np.abs(g - refp).max() / np.abs(refp).max()
        wl = np.array([L['G'], L['D_KL'], L['logP']])
        e_l = (np.abs(l3 - wl) / np.maximum(np.abs(wl), 1)).max()
        drift = (np.abs(l3 - want[t]) / np.maximum(np.abs(want[t]), 1)).max()
        # per tensor: gradient error on the tensor's own scale; parameter distance to the float64 run over the float64 move
        worst = sorted(((np.abs(g[off:off + int(np.prod(sh))].reshape(sh) - G[n]).max() / max(np.abs(G[n]).max(), 1e-30), n)
                        for n, (off, sh) in eng.layout.items()), reverse=True)[:2]
        pw = ''
        if TR is not None:
            now = eng.params.cpu().numpy().astype(np.float64)
            dev = sorted(((np.linalg.norm(now[off:off + int(np.prod(sh))] - TR[t][n].ravel()) /
                           max(np.linalg.norm(TR[t][n].ravel() - np.asarray(P0[n], np.float64).ravel()), 1e-30), n)
                          for n, (off, sh) in eng.layout.items()), reverse=True)[:2]
            pw = '  | param dev ' + ', '.join('%s %.1e' % (n.split('/')[-2] + '/' + n.split('/')[-1] if '/' in n else n, e) for e, n in dev)
        print('    worst tensors (own scale): ' + ', '.join('%s %.1e' % (n, e) for e, n in worst) + pw)
        print('%-16s step %2d  G %.3f  g_raw %.2e  g_pin %.2e  flips %d  loss %.2e  drift vs f64 run %.2e'
              % (tag, t, l3[0], e_raw, e_pin, flips, e_l, drift), flush=True)
