"""Developer tool: the default train step at F frames as eager launches vs one hipGraph replay per step (Stepper.capture), interleaved.
usage: python scripts/graph_vs_eager.py [F] [STEPS]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd')); sys.path.insert(0, ROOT)
import torch
from hipvae import Engine
from hipvae.dp import Stepper
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
eng = Engine(arch)
eng.init_params(0)
st = Stepper(eng, 1e-4, 0.5, 0.999)
g = torch.Generator().manual_seed(1)
x = (torch.rand(F, 513, generator=g) * 2 - 1).cuda()
y = torch.randint(0, 10, (F,), generator=g).cuda()
for _ in range(10):
    st.step(x, y)
def run(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3
for rnd in range(3):
    e = run(lambda: st.step(x, y))
    st.capture(x, y)
    for _ in range(5):
        st.replay()
    gr = run(st.replay)
    print('F=%d round %d: eager %.4f ms  graph %.4f ms' % (F, rnd, e, gr))
