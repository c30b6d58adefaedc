"""Host-side mirror of the reference's plugin surface (no GPU needed)."""
import os
import sys
import types

import numpy as np
import pytest

from helpers import PKG
from oracle import convvae_oracle as O


def test_cli_requires_model_and_trainer():
    import main as train_cli
    with pytest.raises(ValueError, match='Both `model` and `trainer`'):
        train_cli.parse_args([])
    a = train_cli.parse_args(['--model', 'ConvVAE', '--trainer', 'VAETrainer'])
    assert a.model_module == 'model.vae' and a.trainer_module == 'trainer.vae'
    import convert as conv_cli
    with pytest.raises(ValueError, match='You MUST specify `model`'):
        conv_cli.parse_args([])
    c = conv_cli.parse_args(['--model', 'ConvVAE'])
    assert c.src == 'SF1' and c.trg == 'TM3' and c.module == 'model.vae'


def test_convert_groups_utterances_and_cuts_the_result_back():
    """convert.batched / convert.convert_utterances (round 5): consecutive utterances share one device launch up to
    --batch_frames frames, order preserved, results cut at the file boundaries; 0 = one launch per file (the reference's
    sess.run per utterance, convert.py:105-116).  A stand-in machine (frame-wise: row -> 2 * row + target id) and an identity
    normaliser keep this on the CPU."""
    import torch
    import convert as conv_cli
    assert conv_cli.parse_args(['--model', 'ConvVAE']).batch_frames == 16384
    rng = np.random.default_rng(0)
    lens = [300, 700, 5000, 20000, 100, 16384, 1]
    feats = [{'sp': rng.standard_normal((n, 513)).astype(np.float32), 'filename': ('%d' % i).encode()} for i, n in enumerate(lens)]
    groups = list(conv_cli.batched(iter(feats), 16384))
    assert [[int(f['filename']) for f in g] for g in groups] == [[0, 1, 2], [3], [4], [5], [6]]
    assert [[int(f['filename']) for f in g] for g in conv_cli.batched(iter(feats), 0)] == [[i] for i in range(len(lens))]
    assert list(conv_cli.batched(iter([]), 16384)) == []
    calls = []

    class Machine(object):
        def encode(self, x):
            calls.append(int(x.shape[0]))
            assert tuple(x.shape[1:]) == (1, 513, 1)
            return x.reshape(x.shape[0], -1)

        def decode(self, z, y):
            assert y.dtype == torch.int64 and tuple(y.shape) == (z.shape[0],)
            return (2 * z + y.to(z.dtype).unsqueeze(1)).reshape(z.shape[0], 513, 1, 1)

    class Ident(object):
        def forward_process(self, x):
            return torch.as_tensor(np.asarray(x))

        def backward_process(self, x):
            return x

    outs = []
    for g in groups:
        outs += conv_cli.convert_utterances(Machine(), Ident(), [f['sp'] for f in g], 9)
    assert calls == [6000, 20000, 100, 16384, 1]
    for f, o in zip(feats, outs):
        assert tuple(o.shape) == f['sp'].shape and np.array_equal(o.numpy(), 2 * f['sp'] + 9)
    assert conv_cli.convert_utterances(Machine(), Ident(), [], 9) == []


def test_plugin_lookup_by_name():
    from importlib import import_module
    assert hasattr(import_module('model.vae'), 'ConvVAE')
    assert hasattr(import_module('trainer.vae'), 'VAETrainer')


def test_speaker_index_plumbing():
    import analyzer
    assert analyzer.SPEAKERS == ['SF1', 'SF2', 'SF3', 'SM1', 'SM3', 'TF1', 'TF2', 'TM1', 'TM2', 'TM3']
    assert analyzer.SPEAKERS.index('TM3') == 9 and analyzer.SPEAKERS.index('SF1') == 0
    assert analyzer.FEAT_DIM == 1029 and analyzer.RECORD_BYTES == 4116


def test_validate_log_dirs(tmp_path):
    from util.wrapper import validate_log_dirs
    ns = types.SimpleNamespace(logdir_root=None, logdir=None, restore_from=None)
    d = validate_log_dirs(ns)
    assert d['logdir_root'] == 'logdir' and d['logdir'].startswith(os.path.join('logdir', 'train'))
    assert d['restore_from'] == d['logdir']
    import re
    assert re.match(r'\d{4}-\d{4}-\d{2}-\d{4}$', os.path.basename(d['logdir']))
    with pytest.raises(ValueError):
        validate_log_dirs(types.SimpleNamespace(logdir_root=None, logdir='a', restore_from='b'))
    d = validate_log_dirs(types.SimpleNamespace(logdir_root=str(tmp_path), logdir=None, restore_from=None))
    assert d['logdir'].startswith(str(tmp_path))


def test_convert_f0_matches_oracle(tmp_path):
    import convert as conv_cli
    np.array([5.0, 0.2], np.float32).tofile(tmp_path / 'SF1.npf')
    np.array([4.8, 0.3], np.float32).tofile(tmp_path / 'TM3.npf')
    f0 = np.array([0.0, 80.0, 100.0, 250.0, 0.5, 1.0, 2.0], np.float32)
    got = conv_cli.convert_f0(f0, 'SF1', 'TM3', etc_dir=str(tmp_path))
    want = O.convert_f0(f0, 5.0, 0.2, 4.8, 0.3)
    assert got.dtype == np.float32 and np.allclose(got, want, rtol=1e-6)


def test_status_message_format():
    from trainer.vae import VAETrainer
    msg = VAETrainer._status_message(None, 42, -644.377, 87.2)
    assert msg == 'Iter 00042: log P(x|z, y) = -6.444e+02 D_KL(z) = 8.720e+01 '


def test_npf_dtype_trap(tmp_path):
    import analyzer
    v = np.linspace(-10, -2, 513)
    v.astype(np.float32).tofile(tmp_path / 'a.npf')
    v.astype(np.float64).tofile(tmp_path / 'b.npf')
    a = analyzer.load_npf(str(tmp_path / 'a.npf'))
    b = analyzer.load_npf(str(tmp_path / 'b.npf'))
    assert a.dtype == np.float32 and b.dtype == np.float32 and np.allclose(a, b)
    v[:100].astype(np.float32).tofile(tmp_path / 'c.npf')
    with pytest.raises(ValueError):
        analyzer.load_npf(str(tmp_path / 'c.npf'))


def test_engine_fails_loudly_without_gpu(arch):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from hipvae import Engine, HipVaeError
    with pytest.raises(HipVaeError, match='no CPU implementation'):
        Engine(arch)


def test_product_does_not_import_oracle():
    bad = []
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(root, f)).read()
                if 'oracle' in src.replace('no CPU fallback', ''):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_shard_range():
    from hipvae.dp import shard_range
    assert shard_range(2048, 3, 8) == (768, 1024)
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


def test_bounded_shuffler_semantics():
    """analyzer.BoundedShuffler = string_input_producer (file order reshuffled per epoch, records in order) +
    shuffle_batch (pool of `capacity`, at least `min_after_dequeue` left after a draw; analyzer.py:103-135,
    main.py:65-66): every record exactly once per epoch, displacement bounded by the pool, deterministic."""
    import analyzer
    sizes = [700, 300, 1000, 48]
    n = sum(sizes)
    sh = analyzer.BoundedShuffler(sizes, capacity=256, min_after_dequeue=128, seed=7)
    got = np.concatenate([sh.next(16) for _ in range(2 * n // 16)])
    assert got.shape == (2 * n,)
    # a record can only be drawn while it sits in the pool, which holds the next <= 256 entries of the stream:
    # the k-th draw comes from stream positions < k + 256, so every prefix of the output is nearly an epoch prefix
    first = got[:n]
    counts = np.bincount(first, minlength=n)
    assert counts.max() <= 2 and (counts == 0).sum() <= 256           # at most one pool short of a full epoch
    assert np.bincount(got, minlength=n).min() >= 1                    # two epochs cover everything
    # locality: rebuild the stream (same seed => same file orders) and check the displacement bound
    ref = analyzer.BoundedShuffler(sizes, capacity=256, min_after_dequeue=128, seed=7)
    rng_stream = ref._more(3 * n)
    pos = {}
    for i, r in enumerate(rng_stream):
        pos.setdefault(int(r), []).append(i)
    used = {k: 0 for k in pos}
    for k, r in enumerate(got):
        p = pos[int(r)][used[int(r)]]
        used[int(r)] += 1
        assert p < k + 256 + 16, (k, p)
    # the draws really are shuffled, and the sequence is a pure function of the seed
    assert np.abs(np.diff(got[:200])).max() > 1
    again = analyzer.BoundedShuffler(sizes, capacity=256, min_after_dequeue=128, seed=7)
    assert np.array_equal(np.concatenate([again.next(16) for _ in range(2 * n // 16)]), got)
    other = analyzer.BoundedShuffler(sizes, capacity=256, min_after_dequeue=128, seed=8)
    assert not np.array_equal(other.next(64), got[:64])
    # batches larger than capacity - min_after_dequeue are served in several draws
    big = analyzer.BoundedShuffler(sizes, capacity=256, min_after_dequeue=128, seed=1).next(1000)
    assert big.shape == (1000,) and len(set(big.tolist())) == 1000
    with pytest.raises(ValueError):
        analyzer.BoundedShuffler(sizes, capacity=128, min_after_dequeue=128)


def test_bin_writer_roundtrip(tmp_path):
    """analyzer.write_bin writes the reference's record layout (analyzer.py:39-72; README 'Binary data format'):
    read_whole_features and the oracle's parser both read it back."""
    import analyzer
    rng = np.random.default_rng(0)
    n = 37
    sp, ap = rng.standard_normal((n, 513)).astype(np.float32), rng.random((n, 513)).astype(np.float32)
    f0, en = rng.uniform(0, 300, n).astype(np.float32), rng.random(n).astype(np.float32)
    path = str(tmp_path / 'Training Set' / 'TM3' / '100001.bin')
    rows = analyzer.write_bin(path, sp, ap, f0, en, 'TM3')
    assert os.path.getsize(path) == n * analyzer.RECORD_BYTES and rows.shape == (n, 1029)
    feat = next(analyzer.read_whole_features(path))
    assert np.array_equal(feat['sp'], sp) and np.array_equal(feat['ap'], ap)
    assert np.array_equal(feat['f0'], f0) and np.array_equal(feat['en'], en)
    assert feat['speaker'].dtype == np.int64 and set(feat['speaker'].tolist()) == {9}
    want = O.parse_records(open(path, 'rb').read())
    for k in ('sp', 'ap', 'f0', 'en', 'speaker'):
        assert np.array_equal(feat[k], want[k])
    with pytest.raises(ValueError):
        analyzer.write_bin(path, sp[:, :100], ap, f0, en, 0)


def test_pw2wav_inputs_match_reference_arithmetic():
    """analyzer.pw2wav hands float64 C-contiguous arrays to pyworld.synthesize, with 10^sp * en evaluated in the
    arrays' own dtype for the dict form (analyzer.py:160-171) and in float64 for the matrix form (172-185)."""
    import analyzer
    rng = np.random.default_rng(1)
    n = 11
    sp = rng.uniform(-6, -1, (n, 513)).astype(np.float32)
    ap, f0 = rng.random((n, 513)).astype(np.float32), rng.uniform(0, 300, n).astype(np.float32)
    en = rng.uniform(1e-3, 1, n).astype(np.float32)
    got = analyzer.pw2wav_inputs({'sp': sp, 'ap': ap, 'f0': f0, 'en': en})
    want = O.pw2wav_inputs(sp, ap, f0, en)
    for g, w in zip(got, want):
        assert g.dtype == np.float64 and g.flags['C_CONTIGUOUS'] and np.array_equal(g, w)
    assert np.array_equal(got[1], (en.reshape(-1, 1) * np.power(np.float32(10.), sp)).astype(np.float64))
    mat = np.concatenate([sp, ap, f0[:, None], en[:, None]], 1)
    f0m, spm, apm = analyzer.pw2wav_inputs(mat)
    assert np.allclose(spm, en.astype(np.float64).reshape(-1, 1) * 10.0 ** sp.astype(np.float64), rtol=1e-12)
    assert np.array_equal(f0m, f0.astype(np.float64)) and np.array_equal(apm, ap.astype(np.float64))


def test_event_file_writer_roundtrip(tmp_path):
    """util.summary writes TensorBoard event files by hand (TFRecord framing with masked CRC-32C, Event/Summary/
    HistogramProto wire format); read_events parses them back and checks both CRCs."""
    from util import summary as S
    assert S.crc32c(b'123456789') == 0xE3069283                      # CRC-32C check value (RFC 3720)
    lim = S.default_bucket_limits()
    assert len(lim) == 1551 and lim[775] == 0.0 and lim[776] == 1e-12 and np.all(np.diff(lim) > 0)
    w = S.EventWriter(str(tmp_path))
    v = np.array([-1.0, -0.5, 0.0, 0.25, 0.25, 3.0])
    counts = np.bincount(np.searchsorted(lim, v, side='right'), minlength=len(lim) + 1)
    h = S.histogram_proto([v.min(), v.max(), v.sum(), (v ** 2).sum()], counts, lim)
    w.add(12, scalars=[('KL-div', 1.5), ('logPx', -644.25)], histograms=[('x', h)])
    w.close()
    ev = S.read_events(w.path)
    assert ev[0]['file_version'] == 'brain.Event:2' and ev[1]['step'] == 12
    assert ev[1]['scalars'] == {'KL-div': 1.5, 'logPx': -644.25}
    hh = ev[1]['histograms']['x']
    assert hh['min'] == -1.0 and hh['max'] == 3.0 and hh['num'] == 6 and hh['sum'] == v.sum()
    assert hh['bucket'].sum() == 6 and len(hh['bucket']) == len(hh['bucket_limit'])
    # every value falls into the bucket whose limit is the first one above it
    for val in v:
        i = np.searchsorted(hh['bucket_limit'], val, side='right')
        assert hh['bucket'][i] >= 1


def test_rank_seeds_differ():
    from hipvae.dp import rank_seed
    assert len({rank_seed(s, r, 8) for s in range(4) for r in range(8)}) == 32


def test_tf_checkpoint_bundle_roundtrip_and_import(tmp_path):
    """util.tf_checkpoint: TensorFlow V2 checkpoints (LevelDB-format index + raw data shard) written and read without
    TensorFlow; the reference's variable names map onto the flat parameter buffer (parameters, Adam slots, step)."""
    import struct
    import torch
    from helpers import SMALL_ARCH
    from util import tf_checkpoint as T
    rng = np.random.default_rng(0)
    tensors = {'layer_%02d/kernel' % i: rng.standard_normal((3, 1, i + 1, 2)).astype(np.float32) for i in range(40)}
    tensors['global_step'] = np.array(1234, np.int64)
    tensors['counts'] = np.arange(7, dtype=np.int32)
    prefix = str(tmp_path / 'model.ckpt-1234')
    T.write_bundle(prefix, tensors)
    raw = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) > 48
    # prefix compression really happened (41 keys share 'layer_'): the index is much smaller than the sum of key lengths + entries
    got = T.read_bundle(prefix)
    assert set(got) == set(tensors)
    for k in tensors:
        assert got[k].dtype == tensors[k].dtype and got[k].shape == tensors[k].shape and np.array_equal(got[k], tensors[k])
    # corruption is detected: a flipped byte in the data shard (tensor CRC) and in the index (block CRC)
    d = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    d[10] ^= 0xFF
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(d))
    with pytest.raises(ValueError, match='CRC'):
        T.read_bundle(prefix)
    T.write_bundle(prefix, tensors)
    b = bytearray(raw)
    b[20] ^= 0xFF
    open(prefix + '.index', 'wb').write(bytes(b))
    with pytest.raises(ValueError, match='CRC'):
        T.read_bundle(prefix)
    open(prefix + '.index', 'wb').write(raw[:-1] + b'\x00')
    with pytest.raises(ValueError, match='magic'):
        T.read_bundle(prefix)
    # the reference's variables <-> the flat buffer (names = SURVEY App. A.5 = vaenpvc_param_info)
    lay = O.param_layout(SMALL_ARCH)
    layout, off = {}, 0
    for name, shp in lay.items():
        layout[name] = (off, tuple(shp))
        off += int(np.prod(shp))
    state = {'params': torch.tensor(rng.standard_normal(off).astype(np.float32)), 'm': torch.tensor(rng.standard_normal(off).astype(np.float32)),
             'v': torch.tensor(rng.random(off).astype(np.float32)), 'step': 77}
    p2 = str(tmp_path / 'tf' / 'model.ckpt-77')
    T.export_checkpoint(p2, layout, state)
    names = T.read_bundle(p2)
    # names as the reference's graph creates them: conv2d_nchw_layernorm opens variable_scope(name) AND names the conv layer
    # `name` (util/layers.py:55-64), so the conv variables are doubly scoped while the LayerNorm pair is not (layers.py:65)
    assert 'Encoder/Conv2d-0/Conv2d-0/kernel' in names and 'Encoder/Conv2d-0/Conv2d-0/kernel/Adam_1' in names
    assert 'Encoder/Conv2d-1/Conv2d-1/bias' in names and 'Encoder/Conv2d-0/kernel' not in names
    assert 'Encoder/Conv2d-0/layernorm.scale' in names and 'Generator/conv2d_transpose_1/kernel' in names
    assert int(names['global_step']) == 77 and names['global_step'].dtype == np.int32      # tf.Variable(0) is int32
    assert names['Generator/fully_connected/weights'].shape == tuple(lay['Generator/fully_connected/weights'])
    back = T.import_checkpoint(p2, layout)
    for k in ('params', 'm', 'v'):
        assert torch.equal(back[k], state[k])
    assert back['step'] == 77
    # util.wrapper finds and reads it like one of its own checkpoints
    from util.wrapper import find_ckpt, read_ckpt
    assert find_ckpt(str(tmp_path / 'tf')) == p2
    assert torch.equal(read_ckpt(p2, layout)['params'], state['params'])
    # a checkpoint of another model is refused by name
    del names['Encoder/dense/bias']
    T.write_bundle(p2, names)
    with pytest.raises(KeyError, match='lacks'):
        T.import_checkpoint(p2, layout)
    # a kernel stored transposed (same element count, other shape) is refused instead of loaded silently
    p3 = str(tmp_path / 'tf3' / 'model.ckpt-77')
    T.export_checkpoint(p3, layout, state)
    names = T.read_bundle(p3)
    k0 = 'Encoder/Conv2d-1/Conv2d-1/kernel'
    names[k0] = np.ascontiguousarray(np.swapaxes(names[k0], 2, 3))
    T.write_bundle(p3, names)
    with pytest.raises(ValueError, match='shape'):
        T.import_checkpoint(p3, layout)
    # parameters without Adam slots (a Saver over the trainable variables only): the optimiser restarts at step 0
    p4 = str(tmp_path / 'tf4' / 'model.ckpt-77')
    T.export_checkpoint(p4, layout, state)
    names = {k: v for k, v in T.read_bundle(p4).items() if not (k.endswith('/Adam') or k.endswith('/Adam_1'))}
    T.write_bundle(p4, names)
    with pytest.warns(UserWarning, match='no Adam slots'):
        back = T.import_checkpoint(p4, layout)
    assert back['step'] == 0 and torch.equal(back['params'], state['params']) and float(back['m'].abs().max()) == 0.0


def test_bounded_shuffler_state_roundtrip():
    """A checkpointed shuffler continues the SAME record sequence (round-2 advisor: a restored run replayed the first
    batches of the original one)."""
    from analyzer import BoundedShuffler
    a = BoundedShuffler([50, 70, 30], capacity=64, min_after_dequeue=32, seed=5)
    for _ in range(7):
        a.next(16)
    import io
    import torch
    buf = io.BytesIO()
    torch.save({'source': {'shuffler': a.state_dict()}}, buf)
    buf.seek(0)
    sd = torch.load(buf, map_location='cpu')['source']['shuffler']     # torch.load's weights_only default accepts it
    want = [a.next(16) for _ in range(20)]
    b = BoundedShuffler([50, 70, 30], capacity=64, min_after_dequeue=32, seed=999)   # another seed: the state decides
    b.load_state_dict(sd)
    got = [b.next(16) for _ in range(20)]
    assert all(np.array_equal(x, y) for x, y in zip(want, got))
    fresh = BoundedShuffler([50, 70, 30], capacity=64, min_after_dequeue=32, seed=5)
    assert not np.array_equal(fresh.next(16), want[0])    # (and it is not simply the start of the stream again)
    with pytest.raises(ValueError):
        BoundedShuffler([50, 70], capacity=64, min_after_dequeue=32).load_state_dict(sd)


def test_index_ahead_same_stream_and_restore():
    """Drawing the index stream a block of batches ahead hands out the SAME batches as one draw per iteration (both ranks'
    slices), and a checkpoint taken in the middle of a block continues exactly where it stopped."""
    import io
    import torch
    from analyzer import BoundedShuffler, IndexAhead

    def mk(seed):
        return BoundedShuffler([50, 70, 30], capacity=64, min_after_dequeue=32, seed=seed)

    def draw(ia, n):
        out = []
        for _ in range(n):
            blk = ia.next_block_if_due()
            if blk is not None:
                cur = blk
            out.append(cur[ia.take()].copy())
        return out, cur

    plain = mk(5)
    want = [plain.next(16) for _ in range(100)]
    for lo, hi in ((0, 8), (8, 16)):
        ia = IndexAhead(mk(5), 16, lo, hi)
        got, cur = draw(ia, 45)                        # 45: in the middle of the second block of 32
        assert all(np.array_equal(w[lo:hi], g) for w, g in zip(want, got))
        buf = io.BytesIO()
        torch.save({'source': ia.state_dict()}, buf)
        buf.seek(0)
        sd = torch.load(buf, map_location='cpu')['source']
        assert sd['consumed'] == 45 - IndexAhead.AHEAD
        ib = IndexAhead(mk(999), 16, lo, hi)
        ib.load_state_dict(sd)
        assert np.array_equal(ib._block, cur)          # the block in flight is drawn again, identically
        rest = [ib._block[ib.take()].copy() for _ in range(IndexAhead.AHEAD - sd['consumed'])]
        more, _ = draw(ib, 100 - 45 - len(rest))
        assert all(np.array_equal(w[lo:hi], g) for w, g in zip(want[45:], rest + more))
    # a state saved before the first draw, and an old checkpoint without the `consumed` field
    ia = IndexAhead(mk(5), 16, 0, 16)
    assert ia.state_dict()['consumed'] == 0
    ib = IndexAhead(mk(1), 16, 0, 16)
    ib.load_state_dict({'shuffler': mk(5).state_dict()})
    got, _ = draw(ib, 3)
    assert all(np.array_equal(w, g) for w, g in zip(want, got))


def test_bench_site_groups_cover_every_tagged_launch_site():
    """bench.py's roofline.sites adds up kernel GROUPS timed through the library's site tags: every tag of every group must be
    a tag the library really uses (a string literal in csrc/), every tag of scripts/site_times.py must belong to exactly one
    group, and no tag may sit in two groups -- otherwise a new kernel silently drops out of (or is counted twice in) the sum
    the bench line reports."""
    import importlib.util
    import re
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    groups = {name: g['tags'].split() for name, g in bench.SITE_GROUPS.items()}
    flat = [t for tags in groups.values() for t in tags]
    assert len(flat) == len(set(flat)), 'a site tag sits in two groups: %s' % sorted(t for t in set(flat) if flat.count(t) > 1)
    src = ''
    csrc = os.path.join(ROOT, 'vae-npvc_amd', 'csrc')
    for fn in os.listdir(csrc):
        if fn.endswith(('.hip', '.h')):
            src += open(os.path.join(csrc, fn)).read()
    literals = set(re.findall(r'"([a-z0-9_]+)"', src))
    missing = [t for t in flat if t not in literals]
    assert not missing, 'tags unknown to the library: %s' % missing
    st = open(os.path.join(ROOT, 'scripts', 'site_times.py')).read()
    listed = set(re.search(r"TAGS = \((.*?)\)\.split\(\)", st, re.S).group(1).replace("'", ' ').split())
    assert listed == set(flat), (sorted(listed - set(flat)), sorted(set(flat) - listed))
    # ... and every tag the layered path's launch code passes to VAENPVC_TIMED is listed
    used = set(re.findall(r'VAENPVC_TIMED\("([a-z0-9_]+)"', src))
    small_batch = {t for t in used if t.startswith('frame_')}       # the small-batch frame kernels (not part of the layered step)
    assert used - small_batch <= set(flat), sorted(used - small_batch - set(flat))
