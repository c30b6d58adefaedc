"""End-to-end rate of the trainer's own hot loop (trainer/vae.py:94-99: dequeue -> sess.run(opt['g'])) through the plugin
surface on a synthetic .bin tree: iterations per second at the reference's batch size, next to the bare step time bench.py
reports for resident inputs.  usage: python scripts/trainer_bench.py [iterations] [batch]"""
import json, os, sys, tempfile, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vae-npvc_amd'))
import numpy as np
import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
import analyzer
from model.vae import ConvVAE
from trainer.vae import VAETrainer

arch = json.load(open(os.path.join(ROOT, 'vae-npvc_amd', 'architecture-vae-vcc2016.json')))
root = tempfile.mkdtemp()
rng = np.random.default_rng(0)
recs = []
for spk_id, spk in [(0, 'SF1'), (9, 'TM3')]:
    d = os.path.join(root, 'bin', 'Training Set', spk)
    os.makedirs(d)
    for u in range(8):
        n = int(rng.integers(400, 800))
        r = rng.standard_normal((n, 1029)).astype(np.float32)
        r[:, :513] = rng.uniform(-12, -3, (n, 513))
        r[:, -1] = spk_id
        r.tofile(os.path.join(d, '1000%02d.bin' % u))
        recs.append(r)
allr = np.concatenate(recs)
xmin = np.percentile(allr[:, :513], 0.5, axis=0).astype(np.float32)
xmax = np.percentile(allr[:, :513], 99.5, axis=0).astype(np.float32)
arch['training']['batch_size'] = B
arch['training']['datadir'] = os.path.join(root, 'bin', 'Training Set', '*', '*.bin')
image, label = analyzer.read(arch['training']['datadir'], B, normalizer=analyzer.Tanhize(xmax=xmax, xmin=xmin), seed=3)
machine = ConvVAE(arch, seed=5)
loss = machine.loss(image, label)
dirs = {'logdir': os.path.join(root, 'logdir', 'train', 'stamp')}
out = {}
for warm, n in ((True, 200), (False, N)):
    arch['training']['max_iter'] = (0 if warm else 200) + n
    tr = VAETrainer(loss, arch, types.SimpleNamespace(seed=17, restore_from=None, ckpt=None), dirs) if warm else tr
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(nIter=0, machine=machine, status_secs=1e9, save_secs=1e9, summary_secs=1e9)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if not warm:
        out = {'batch': B, 'iterations': n, 'ms_per_iteration': dt / n * 1e3, 'frames_per_s': B * n / dt}
print(json.dumps(out))
