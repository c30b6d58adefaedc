"""Experiment: does running two half batches concurrently (two contexts, two streams) beat one full batch?
(HBM-bound and matrix-core-bound kernels of different layers would overlap.)  usage: python scripts/exp_two_halves.py [frames]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd'))
import torch
from hipvae import Engine

F = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
g = torch.Generator().manual_seed(0)


def mk(Fh):
    e = Engine(arch)
    e.init_params(0)
    x = (torch.rand(Fh, 513, generator=g) * 2 - 1).cuda()
    y = torch.randint(0, 10, (Fh,), generator=g).cuda()
    gr = torch.zeros(e.n_params, device='cuda')
    return e, x, y, gr


def run(parts, n):
    streams = [torch.cuda.Stream() for _ in parts]
    def once(i):
        for (e, x, y, gr), s in zip(parts, streams):
            with torch.cuda.stream(s):
                e.train_fwd_bwd(x, y, None, gr, seed=1, offset=i)
    for i in range(3):
        once(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        once(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = [mk(F)]
print('one batch of %d frames : %.3f ms per fwd+bwd' % (F, run(full, 20)))
del full
torch.cuda.empty_cache()
for k in (2, 4):
    parts = [mk(F // k) for _ in range(k)]
    print('%d concurrent batches of %d: %.3f ms per fwd+bwd of all' % (k, F // k, run(parts, 20)))
    del parts
    torch.cuda.empty_cache()
