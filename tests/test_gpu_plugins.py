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
    import types
    trainer = VAETrainer(loss, arch, types.SimpleNamespace(seed=17, restore_from=None, ckpt=None), dirs)
    # the sampler draws on the device (Philox keyed by the trainer's seed, counter = global step): the oracle
    # follows the same trajectory on the NumPy restatement of that draw
    from oracle import philox_ref
    eps_list = [torch.tensor(philox_ref.normal(16 * 128, trainer.opt['g'].seed, t).reshape(16, 128)) for t in range(3)]
    ckpt = trainer.train(nIter=123456, machine=machine, summary_secs=0)     # nIter ignored (trap T7)
    assert os.path.basename(ckpt) == 'model.ckpt-3' and os.path.exists(ckpt)
    assert os.path.exists(os.path.join(dirs['logdir'], 'training.log'))
    # summaries of model/vae.py:132-136 in a TensorBoard event file (written on every step here: summary_secs=0)
    from util.summary import read_events
    evf = [f for f in os.listdir(dirs['logdir']) if f.startswith('events.out.tfevents')]
    assert len(evf) == 1
    ev = read_events(os.path.join(dirs['logdir'], evf[0]))
    assert [e['step'] for e in ev[1:]] == [1, 2, 3]
    assert set(ev[1]['scalars']) == {'KL-div', 'logPx'} and set(ev[1]['histograms']) == {'x', 'xh'}
    hx = ev[1]['histograms']['x']
    assert hx['num'] == 16 * 513 and abs(hx['min'] - batches[0][0].min()) < 1e-6 and abs(hx['max'] - batches[0][0].max()) < 1e-6
    assert abs(hx['sum'] - batches[0][0].astype(np.float64).sum()) < 1e-3
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
        # (a gradient entry carries an error of ~1e-5 of ITS TENSOR's largest entry; at 1e-2 of the global maximum
        #  that is at most ~1e-3 of the entry itself, which is what the normalised Adam update then inherits)
        ok = np.abs(g) > 1e-2 * np.abs(g).max()
        strong = ok if strong is None else (strong & ok)
        p, m, v = O.tf_adam_step(p, g, m, v, t)
    got = machine.engine.params.cpu().numpy().astype(np.float64) - p0
    want = p - p0
    assert strong.sum() > 1000
    assert np.abs(got - want)[strong].max() / np.abs(want).max() < 2e-2    # (three steps: see test_hipgraph_replay_matches_eager)
    assert np.mean(np.abs(got - want) > 0.05 * np.abs(want).max()) < 0.02   # noise-floor entries are rare
    # restore path (util/wrapper.load) and the architecture-next-to-checkpoint contract
    from util.wrapper import load
    m2 = ConvVAE(arch, seed=99)
    step = load(m2.engine, dirs['logdir'], ckpt='model.ckpt-3')
    assert step == 3 and torch.equal(m2.engine.params, machine.engine.params)
    # --restore_from: parameters, Adam slots and global_step come back and training continues from step 3
    arch2 = json.loads(json.dumps(arch))
    arch2['training']['max_iter'] = 5
    m3 = ConvVAE(arch2, seed=123)
    image3, label3 = analyzer.read(arch['training']['datadir'], 16, normalizer=normalizer, seed=3)
    dirs3 = {'logdir': os.path.join(str(tmp_path), 'logdir', 'train', 'stamp2'), 'restore_from': dirs['logdir']}
    tr3 = VAETrainer(m3.loss(image3, label3), arch2, types.SimpleNamespace(seed=17, restore_from=dirs['logdir'], ckpt=None), dirs3)
    ck3 = tr3.train(nIter=0, machine=m3)
    assert os.path.basename(ck3) == 'model.ckpt-5' and tr3.opt['g'].step_count == 5
    assert float(tr3.opt['g'].m.abs().sum()) > 0 and float(tr3.opt['g'].v.abs().sum()) > 0     # Adam slots restored and moving


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


def test_batched_conversion_equals_per_utterance_conversion(tmp_path):
    """convert.convert_utterances: several utterances in one launch (frames are independent samples of the frame-wise network)
    give what one launch per utterance gives, at the conversion bar (1e-4 of the converted spectrum's range; the two runs may
    use different kernel families -- whole-frame kernels up to 512 frames, layered kernels above)."""
    import analyzer
    import convert as conv_cli
    from model.vae import ConvVAE
    arch = load_arch()
    allr, xmin, xmax = make_dataset(str(tmp_path), n_utt=1, seed=9)
    normalizer = analyzer.Tanhize(xmax=xmax, xmin=xmin)
    machine = ConvVAE(arch, seed=2)
    rng = np.random.default_rng(5)
    lens = [300, 37, 1500, 700, 1]
    lo, hi = xmin.astype(np.float32), xmax.astype(np.float32)
    sps = [(lo + (hi - lo) * rng.random((n, 513), dtype=np.float32)).astype(np.float32) for n in lens]
    trg = analyzer.SPEAKERS.index('TM3')
    one = [conv_cli.convert_utterance(machine, normalizer, sp, trg).cpu().numpy() for sp in sps]
    many = [t.cpu().numpy() for t in conv_cli.convert_utterances(machine, normalizer, sps, trg)]
    assert [m.shape for m in many] == [(n, 513) for n in lens]
    for a, b in zip(one, many):
        assert np.abs(a - b).max() <= 1e-4 * np.abs(a).max()
    # ... and against the float64 oracle at the utterance sizes convert.py actually runs (700 / 1 500 frames)
    P = O.unflatten_params(arch, machine.engine.params.cpu().numpy())
    for i in (2, 3):
        x = O.tanhize_forward(sps[i].astype(np.float64), xmin.astype(np.float64), xmax.astype(np.float64))
        R = O.np_forward(arch, P, x, np.full(len(x), trg), None)
        want = O.tanhize_backward(R['xh'], xmin.astype(np.float64), xmax.astype(np.float64))
        assert np.abs(many[i] - want).max() / np.abs(want).max() < 1e-4


def test_convert_cli_end_to_end(tmp_path, monkeypatch):
    """convert.main() on a synthetic tree (convert.py:66-116): checkpoint + architecture lookup, per-utterance device
    path, log-F0 transform, and the arrays handed to the vocoder (analyzer.pw2wav, analyzer.py:160-171) -- WORLD and
    soundfile themselves are external, so stand-ins record what they are given."""
    import sys
    import types
    import analyzer
    import convert as conv_cli
    from model.vae import ConvVAE
    arch = load_arch()
    root = str(tmp_path)
    allr, xmin, xmax = make_dataset(root, n_utt=2, seed=6)
    os.makedirs(os.path.join(root, 'etc'))
    xmin.tofile(os.path.join(root, 'etc', 'xmin.npf'))
    xmax.astype(np.float64).tofile(os.path.join(root, 'etc', 'xmax.npf'))      # both on-disk dtypes (trap T4)
    np.array([5.0, 0.25], np.float32).tofile(os.path.join(root, 'etc', 'SF1.npf'))
    np.array([4.7, 0.30], np.float32).tofile(os.path.join(root, 'etc', 'TM3.npf'))
    logdir = os.path.join(root, 'logdir', 'train', 'stamp')
    os.makedirs(logdir)
    with open(os.path.join(logdir, 'architecture-vae-vcc2016.json'), 'w') as fp:
        json.dump(arch, fp)
    machine = ConvVAE(arch, seed=8)
    torch.save({'params': machine.engine.params.cpu(), 'step': 7}, os.path.join(logdir, 'model.ckpt-7'))
    calls, written = [], []
    fake_pw = types.ModuleType('pyworld')
    fake_pw.synthesize = lambda f0, sp, ap, fs: (calls.append((f0, sp, ap, fs)) or np.zeros(8))
    fake_sf = types.ModuleType('soundfile')
    fake_sf.write = lambda name, y, fs: written.append((name, fs))
    monkeypatch.setitem(sys.modules, 'pyworld', fake_pw)
    monkeypatch.setitem(sys.modules, 'soundfile', fake_sf)
    monkeypatch.chdir(root)
    pattern = os.path.join(root, 'bin', 'Training Set', '{}', '*.bin')
    out_dir = conv_cli.main(['--src', 'SF1', '--trg', 'TM3', '--model', 'ConvVAE', '--checkpoint',
                             os.path.join(logdir, 'model.ckpt-7'), '--output_dir', os.path.join(root, 'logdir'),
                             '--file_pattern', pattern])
    assert len(calls) == 2 and len(written) == 2
    assert sorted(os.path.basename(n) for n, _ in written) == ['SF1-TM3-100000.wav', 'SF1-TM3-100001.wav']
    assert all(os.path.dirname(n) == out_dir and fs == 16000 for n, fs in written)
    assert os.path.normpath(out_dir).startswith(os.path.normpath(os.path.join(root, 'logdir', 'output')))
    P = O.unflatten_params(arch, machine.engine.params.cpu().numpy())
    trg = analyzer.SPEAKERS.index('TM3')
    files = sorted(os.listdir(os.path.join(root, 'bin', 'Training Set', 'SF1')))
    for (f0, sp, ap, fs), f in zip(calls, files):
        raw = O.parse_records(open(os.path.join(root, 'bin', 'Training Set', 'SF1', f), 'rb').read())
        x = O.tanhize_forward(raw['sp'].astype(np.float64), xmin.astype(np.float64), xmax.astype(np.float64))
        R = O.np_forward(arch, P, x, np.full(len(x), trg), None)
        sp_conv = O.tanhize_backward(R['xh'], xmin.astype(np.float64), xmax.astype(np.float64))
        want_f0, want_sp, want_ap = O.pw2wav_inputs(sp_conv.astype(np.float32), raw['ap'],
                                                    O.convert_f0(raw['f0'], 5.0, 0.25, 4.7, 0.30), raw['en'])
        for a in (f0, sp, ap):
            assert a.dtype == np.float64 and a.flags['C_CONTIGUOUS']
        assert fs == 16000 and np.allclose(f0, want_f0, rtol=1e-6) and np.array_equal(ap, want_ap)
        # 10^sp amplifies the 1e-4 parity bar of the log-spectrum by ln(10) * |range|: compare in the log domain
        # (the synthetic `en` column is signed; signs must agree exactly, magnitudes in the log domain)
        assert np.array_equal(np.sign(sp), np.sign(want_sp))
        la, lb = np.log10(np.abs(sp)), np.log10(np.abs(want_sp))
        assert np.abs(la - lb).max() < 1e-4 * np.abs(lb).max()
