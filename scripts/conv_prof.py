"""Developer tool: run a few train steps with an instrumented variant build
(scripts/build_variant.sh prof "-DVAENPVC_PROF=1", VAENPVC_LIB=variants/prof/libvaenpvc_hip.so)
and print the per-phase cycle shares of every k_convgemm instance."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd'))
import numpy as np
import torch
from hipvae.engine import Engine
from hipvae import lib as L
import json

arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
eng = Engine(arch)
g = torch.Generator(device='cpu').manual_seed(0)
x = (torch.rand(F, 513, generator=g) * 2 - 1).cuda()
y = torch.randint(0, 10, (F,), generator=g).cuda()
eps = torch.randn(F, 128, generator=g).cuda()
grads = torch.zeros(eng.n_params, dtype=torch.float32, device='cuda')
lib = L.load_library()
fn = lib.vaenpvc_debug_conv_prof
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 464)()
for i in range(2):
    eng.train_fwd_bwd(x, y, eps, grads)
fn(None, 1)
N = 3
for i in range(N):
    eng.train_fwd_bwd(x, y, eps, grads)
fn(buf, 0)
a = np.array(buf[:320], dtype=np.float64).reshape(32, 10)
wg = np.array(buf[320:448], dtype=np.float64).reshape(16, 8)
tb = np.array(buf[448:456], dtype=np.float64)
names = ['gload', 'setup', 'kloop', 'epi', 'bar1', 'lstore', 'bar2']
print('slot waves   total_cyc/wave | ' + ' '.join('%7s' % n for n in names) + ' | other')
for s in range(32):
    if a[s, 0] == 0:
        continue
    w = a[s, 0]
    tot = a[s, 8] / w
    parts = a[s, 1:8] / w
    print('%4d %6d %12.0f | ' % (s, w / N, tot) + ' '.join('%6.1f%%' % (100 * p / tot) for p in parts) + ' | %5.1f%%' % (100 * (tot - parts.sum()) / tot))

names = ['gload', 'bar1', 'compute', 'landing', 'epilogue', 'lstore+l']
print('convwgrad slot waves total_cyc/wave | ' + ' '.join('%8s' % n for n in names) + ' | other')
for s in range(16):
    if wg[s, 0] == 0:
        continue
    w = wg[s, 0]
    tot = wg[s, 6] / w
    parts = wg[s, [1, 2, 3, 4, 5, 7]] / w
    print('%4d %6d %12.0f | ' % (s, w / N, tot) + ' '.join('%7.1f%%' % (100 * p / tot) for p in parts) + ' | %5.1f%%' % (100 * (tot - parts.sum()) / tot))

if tb[0] > 0:
    w = tb[0]
    tot = tb[5] / w
    print('toep_dgrad_bf16: waves %d total_cyc/wave %.0f | tapcopy %.1f%% stage+bar %.1f%% kloop %.1f%% epilogue %.1f%% other %.1f%%' % (
        w / N, tot, 100 * tb[1] / w / tot, 100 * tb[2] / w / tot, 100 * tb[3] / w / tot, 100 * tb[4] / w / tot,
        100 * (tot - tb[1:5].sum() / w) / tot))

tw = np.array(buf[456:464], dtype=np.float64)
if tw[0] > 0:
    w = tw[0]
    tot = tw[7] / w
    print('toep_wgrad_bf16: waves %d total_cyc/wave %.0f | gload %.1f%% loadF %.1f%% mfma %.1f%% lstore %.1f%% barrier %.1f%% epilogue %.1f%%' % (
        w / N, tot, *[100 * tw[i] / w / tot for i in range(1, 7)]))
