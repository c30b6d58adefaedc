"""Soak check of the small-batch path: N Adam steps on one fixed 16-frame batch with the frame kernels (default) and with
the layered kernels (mask bit 21 cleared), same seeds: the two loss trajectories must stay together (a race in the phase
kernels or in the atomics would show as a drift), losses finite and falling.  usage: python scripts/soak_small_batch.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd'))
import torch
from hipvae import Engine
from hipvae.dp import Stepper

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
g = torch.Generator().manual_seed(0)
F = 16
x = torch.tanh(torch.randn(F, 513, generator=g)).cuda()
y = torch.randint(0, 10, (F,), generator=g).cuda()
eps = torch.randn(F, 128, generator=g).cuda()
traj = {}
for name, mask in (('frame', 0xffffffff), ('layered', 0xffffffff & ~(1 << 21))):
    eng = Engine(arch)
    eng.init_params(0)
    eng.set_tuned_masks(mask, mask)
    st = Stepper(eng, 1e-4, 0.5, 0.999)
    out = []
    for i in range(N):
        l3 = st.step(x, y, eps)
        if i % (N // 10) == 0 or i == N - 1:
            out.append([float(v) for v in l3.cpu()])
    traj[name] = out
    print(name, ' '.join('%.3f' % o[0] for o in out))
a, b = torch.tensor(traj['frame']), torch.tensor(traj['layered'])
assert torch.isfinite(a).all() and torch.isfinite(b).all()
rel = ((a - b).abs() / b.abs().clamp_min(1.0)).max().item()
print('largest relative gap between the trajectories: %.2e' % rel)
assert a[-1, 0] < a[0, 0] and rel < 2e-2
print('soak ok')
