"""Duration of every segment of the one-launch weight gradient (gfx950_frame_wgrad.h) run ALONE, and of the whole launch.
usage: python scripts/wgrad_prof.py [frames]"""
import ctypes as C
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd'))
import torch
from hipvae import Engine
from hipvae import lib as L

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
eng = Engine(arch)
eng.init_params(0)
g = torch.Generator().manual_seed(0)
x = (torch.rand(F, 513, generator=g) * 2 - 1).cuda()
y = torch.randint(0, 10, (F,), generator=g).cuda()
eps = torch.randn(F, 128, generator=g).cuda()
grads = torch.zeros(eng.n_params, device='cuda')
lib = L.load_library()
lib.vaenpvc_debug_wg_segments.argtypes = [C.c_uint]
NAMES = ['toeplitz', 'enc4', 'dec0', 'enc3', 'enc2', 'enc1', 'dec1', 'dec2', 'enc0', 'dWz', 'dWy', 'dWmu', 'dWlv', 'embedding',
         'merge biases', 'head biases', 'bias d3', 'LN sums']


def timed(mask, tag='frame_wgrad', n=20):
    lib.vaenpvc_debug_wg_segments(mask)
    for _ in range(3):
        eng.train_fwd_bwd(x, y, eps, grads)
    eng.timer_select(tag)
    for _ in range(n):
        eng.train_fwd_bwd(x, y, eps, grads)
    ms, k = eng.timer_read()
    eng.timer_select(None)
    return ms / max(k, 1) * 1e3


print('frames %d' % F)
print('  %-14s %8.1f us' % ('ALL', timed(0xffffffff)))
for i, nm in enumerate(NAMES):
    print('  %-14s %8.1f us' % (nm, timed(1 << i)))
for tag in ('frame_fwd', 'frame_bwd'):
    print('  %-14s %8.1f us' % (tag, timed(0xffffffff, tag)))

# chunk caps (order: toeplitz, enc4, dec0, enc3, enc2, enc1, dec1, dec2, enc0): whole launch under a few settings
lib.vaenpvc_debug_wg_caps.argtypes = [C.POINTER(C.c_int)]
for caps in ([32, 16, 16, 16, 16, 32, 32, 32, 32], [64, 16, 64, 16, 16, 32, 32, 32, 32], [128, 16, 64, 16, 16, 32, 32, 32, 32],
             [64, 32, 64, 32, 32, 32, 32, 32, 32], [64, 32, 64, 32, 32, 64, 64, 64, 64], [128, 32, 128, 32, 32, 64, 64, 64, 64],
             [64, 8, 32, 8, 8, 16, 16, 16, 16]):
    arr = (C.c_int * 9)(*caps)
    lib.vaenpvc_debug_wg_caps(arr)
    print('  caps %-44s ALL %8.1f us   toeplitz %8.1f us' % (caps, timed(0xffffffff), timed(1)))
lib.vaenpvc_debug_wg_caps(None)
