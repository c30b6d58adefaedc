"""Throughput of the conversion path (convert.py:60-63,79-89): encode(x) -> z_mu, decode(z_mu, target speaker)
on frames resident in HBM.  Usage: python scripts/bench_convert.py [frames] [iterations]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd'))
import torch
from hipvae.engine import Engine

arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = Engine(arch)
eng.init_params(0)
g = torch.Generator().manual_seed(0)
x = (torch.rand(F, 513, generator=g) * 2 - 1).cuda()
y = torch.full((F,), 9, dtype=torch.int64).cuda()
for _ in range(3):
    xh = eng.decode(eng.encode(x), y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    xh = eng.decode(eng.encode(x), y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(json.dumps({'path': 'encode+decode (conversion)', 'frames': F, 'ms': dt * 1e3, 'frames_per_s': F / dt,
                  'algorithmic_tflops': F * 9.419e6 / dt / 1e12}))
