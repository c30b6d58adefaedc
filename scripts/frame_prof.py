"""Per-phase shader-clock profile of the small-batch frame kernels (block 0), VAENPVC_FRAME_PROF=1.
usage: VAENPVC_FRAME_PROF=1 python scripts/frame_prof.py [frames]"""
import ctypes as C
import json
import os
import sys
os.environ['VAENPVC_FRAME_PROF'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd'))
import numpy as np
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
for _ in range(5):
    eng.train_fwd_bwd(x, y, eps, grads)
torch.cuda.synchronize()
lib = L.load_library()
buf = (C.c_longlong * 1024)()
lib.vaenpvc_debug_frame_prof.restype = C.c_int
assert lib.vaenpvc_debug_frame_prof(buf) == 0
a = np.array(buf[:], np.int64).reshape(2, 512)
T3 = ['rsum', 'var', 'apply']
fwd = ['prologue', 'halo_x']
for l in ('e0', 'e1', 'e2', 'e3', 'e4'):
    fwd += [l + '_part'] + [l + '_' + t for t in T3]
fwd += ['heads_part', 'heads_reduce', 'sampler', 'merge_part', 'h_reduce', 'h_halo']
for l in ('d0', 'd1', 'd2'):
    fwd += [l + '_part'] + [l + '_' + t for t in T3]
fwd += ['d3_part', 'xh_nll', 'nll_1']
LB = ['lnb_sums', 'lnb_da']
bwd = ['prologue', 'dxh_stage', 'd3g_part', 'd3g_rload'] + ['d2_' + t for t in LB]
for l, nxt in (('d2g', 'd1'), ('d1g', 'd0')):
    bwd += [l + '_halo', l + '_part', l + '_rload'] + [nxt + '_' + t for t in LB]
bwd += ['d0g_halo', 'd0g_part', 'd0g_reduce', 'mergeG_part', 'reparam', 'headsG_part', 'headsG_rload'] + ['e4_' + t for t in LB]
for l, nxt in (('e4g', 'e3'), ('e3g', 'e2'), ('e2g', 'e1'), ('e1g', 'e0')):
    bwd += [l + '_halo', l + '_part', l + '_rload'] + [nxt + '_' + t for t in LB]
bwd += ['flush_e0']
for name, row, labels in (('forward', a[0], fwd), ('backward', a[1], bwd)):
    n = int(np.count_nonzero(row)) - 1
    d = np.diff(row[:n + 1])
    print('%s: %d phases, %d clocks in all' % (name, n, int(row[n] - row[0])))
    per = len(labels) - 1                      # phases per frame (the prologue runs once)
    if n > 1 + per:                            # several frames per workgroup: totals per frame (is the second one cheaper?)
        for k in range((n - 1) // per):
            print('  frame %d of block 0: %d clocks' % (k, int(d[1 + k * per:1 + (k + 1) * per].sum())))
    agg = {}
    for i in range(n):
        lab = labels[i] if i < len(labels) else 'phase%d' % i
        print('  %-16s %8d' % (lab, d[i]))
        key = lab.split('_', 1)[1] if '_' in lab else lab
        agg[key] = agg.get(key, 0) + int(d[i])
    print('  by kind:', sorted(agg.items(), key=lambda kv: -kv[1]))
