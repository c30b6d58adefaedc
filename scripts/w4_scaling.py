"""Developer: duration of the 1025-tap weight gradient (site dec3_wgrad) at several batch sizes with the same grid (256
workgroups from 4 096 frames on): the difference between two sizes is pure main-loop time, the rest prologue + epilogue.
usage: python scripts/w4_scaling.py [frames ...]   (W4_TAG=enc4_wgrad selects another site)"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd'))
import torch
from hipvae import Engine

TAG = os.environ.get('W4_TAG', 'dec3_wgrad')
arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
eng = Engine(arch)
eng.init_params(0)
eng.set_tuned_masks(0xffffffff, 0xffffffff & ~(1 << 30))      # weight gradients on the caller's stream: serialised kernels
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768]
g = torch.Generator().manual_seed(0)
for F in sizes:
    x = (torch.rand(F, 513, generator=g) * 2 - 1).cuda()
    y = torch.randint(0, 10, (F,), generator=g).cuda()
    eps = torch.randn(F, 128, generator=g).cuda()
    grads = torch.zeros(eng.n_params, device='cuda')
    for _ in range(3):
        eng.train_fwd_bwd(x, y, eps, grads)
    eng.timer_select(TAG)
    for _ in range(10):
        eng.train_fwd_bwd(x, y, eps, grads)
    ms, k = eng.timer_read()
    eng.timer_select(None)
    print('frames %6d  %s %.1f us  (%d launches)' % (F, TAG, ms / max(k, 1) * 1e3, k))
