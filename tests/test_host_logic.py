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
