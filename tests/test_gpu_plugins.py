"""End-to-end through the reference's plugin surface on synthetic .bin data:
main.py wiring (read -> ConvVAE.loss -> VAETrainer.train) and the convert.py tensor path."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import load_arch
from oracle import convvae_oracle as O

pytestmark = pytest.mark.gpu


def make_dataset(root, n_utt=3, seed=0):
    rng = np.random.default_rng(seed)
    recs = []
    for spk_id, spk in [(0, 'SF1'), (9, 'TM3')]:
        d = os.path.join(root, 'bin', 'Training Set', spk)
        os.makedirs(d)
        for u in range(n_utt):
            n = int(rng.integers(40, 80))
            r = rng.standard_normal((n, 1029)).astype(np.float32)
            r[:, :513] = rng.uniform(-12, -3, (n, 513))
            r[:, 1026] = np.where(rng.random(n) > 0.3, rng.uniform(80, 300, n), 0.0)
            r[:, -1] = spk_id
            r.tofile(os.path.join(d, '1000%02d.bin' % u))
            recs.append(r)
    allr = np.concatenate(recs)
    xmin = np.percentile(allr[:, :513], 0.5, axis=0).astype(np.float32)
    xmax = np.percentile(allr[:, :513], 99.5, axis=0).astype(np.float32)
    return allr, xmin, xmax


def test_train_three_steps_through_plugins(tmp_path):
    import analyzer
    from model.vae import ConvVAE
    from trainer.vae import VAETrainer
    arch = load_arch()
    arch['training']['max_iter'] = 3
    arch['training']['batch_size'] = 16
    allr, xmin, xmax = make_dataset(str(tmp_path))
    arch['training']['datadir'] = os.path.join(str(tmp_path), 'bin', 'Training Set', '*', '*.bin')
    normalizer = analyzer.Tanhize(xmax=xmax, xmin=xmin)
    image, label = analyzer.read(arch['training']['datadir'], 16, normalizer=normalizer, seed=3)
    machine = ConvVAE(arch, seed=5)
    p0 = machine.engine.params.cpu().numpy().copy()
    # inject the sampler noise so the oracle can follow the same trajectory
    eps_list = [torch.randn(16, 128, generator=torch.Generator().manual_seed(100 + i)) for i in range(3)]
    it = iter(eps_list)
    machine._draw_eps = lambda F: next(it).to(machine.engine.device)
    batches = []
    orig = image.source.next_batch
    def spy():
        x, y = orig()
        batches.append((x.cpu().numpy().reshape(16, 513), y.cpu().numpy()))
        return x, y
    image.source.next_batch = spy
    loss = machine.loss(image, label)
    assert set(loss.keys()) == {'G', 'D_KL', 'logP'}
    dirs = {'logdir': os.path.join(str(tmp_path), 'logdir', 'train', 'stamp')}
    trainer = VAETrainer(loss, arch, None, dirs)
    ckpt = trainer.train(nIter=123456, machine=machine)          # nIter ignored (trap T7)
    assert os.path.basename(ckpt) == 'model.ckpt-3' and os.path.exists(ckpt)
    assert os.path.exists(os.path.join(dirs['logdir'], 'training.log'))
    # speaker ids are bit-exact and only {0, 9}
    for xb, yb in batches:
        assert yb.dtype == np.int64 and set(yb.tolist()) <= {0, 9}
        assert xb.min() >= -1 and xb.max() <= 1
    # oracle trajectory (float64) on the same batches / eps
    p = p0.astype(np.float64); m = np.zeros_like(p); v = np.zeros_like(p)
    names = list(O.param_layout(arch).keys())
    strong = None
    for t, ((xb, yb), e) in enumerate(zip(batches, eps_list), 1):
        _, G = O.torch_loss_and_grads(arch, O.unflatten_params(arch, p), xb, yb, e.numpy(), torch.float64)
        g = np.concatenate([G[n].ravel() for n in names])
        # Adam's first steps move every weight by ~lr*sign(g): entries whose gradient is at the fp32
        # noise floor have an arbitrary sign, so the trajectory is compared where the gradient is
        # well above that floor in every step.
        ok = np.abs(g) > 1e-3 * np.abs(g).max()
        strong = ok if strong is None else (strong & ok)
        p, m, v = O.tf_adam_step(p, g, m, v, t)
    got = machine.engine.params.cpu().numpy().astype(np.float64) - p0
    want = p - p0
    assert strong.sum() > 1000
    assert np.abs(got - want)[strong].max() / np.abs(want).max() < 5e-3
    assert np.mean(np.abs(got - want) > 0.05 * np.abs(want).max()) < 0.02   # noise-floor entries are rare
    # restore path (util/wrapper.load) and the architecture-next-to-checkpoint contract
    from util.wrapper import load
    m2 = ConvVAE(arch, seed=99)
    step = load(m2.engine, dirs['logdir'], ckpt='model.ckpt-3')
    assert step == 3 and torch.equal(m2.engine.params, machine.engine.params)


def test_convert_utterance_matches_oracle(tmp_path):
    import analyzer
    import convert as conv_cli
    from model.vae import ConvVAE
    arch = load_arch()
    allr, xmin, xmax = make_dataset(str(tmp_path), n_utt=1, seed=4)
    normalizer = analyzer.Tanhize(xmax=xmax, xmin=xmin)
    machine = ConvVAE(arch, seed=1)
    P = O.unflatten_params(arch, machine.engine.params.cpu().numpy())
    f = os.path.join(str(tmp_path), 'bin', 'Training Set', 'SF1', '100000.bin')
    feat = next(analyzer.read_whole_features(f))
    assert feat['speaker'].dtype == np.int64 and set(feat['speaker'].tolist()) == {0}
    trg = analyzer.SPEAKERS.index('TM3')
    got = conv_cli.convert_utterance(machine, normalizer, feat['sp'], trg).cpu().numpy()
    x = O.tanhize_forward(feat['sp'].astype(np.float64), xmin.astype(np.float64), xmax.astype(np.float64))
    R = O.np_forward(arch, P, x, np.full(len(x), trg), None)
    want = O.tanhize_backward(R['xh'], xmin.astype(np.float64), xmax.astype(np.float64))
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-4
    nhwc = machine.decode(machine.encode(torch.tensor(x, dtype=torch.float32, device='cuda').view(-1, 1, 513, 1)),
                          torch.full((len(x),), trg, dtype=torch.int64, device='cuda'))
    assert tuple(nhwc.shape) == (len(x), 513, 1, 1)
