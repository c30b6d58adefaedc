"""world_size-2 / -4 data-parallel host logic under gloo on CPU.

The HIP engine needs a GPU, so the Stepper (hipvae/dp.py) and the trainer loop are driven with a CPU
stand-in backend built from the ORACLE (tests/dp_gloo_worker.py): the thing under test is the sharding /
bucketed all-reduce / grad_scale / broadcast / logging logic, not the arithmetic.
Property: N-rank parameters after k steps == 1-rank parameters on the concatenated batch.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import SMALL_ARCH
from oracle import convvae_oracle as O

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dp_gloo_worker.py')
_port = [29000 + os.getpid() % 2000]


def launch(world, F, steps, out, mode='bucket'):
    _port[0] += 1
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(_port[0]), str(F), str(steps), out, mode])
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


@pytest.mark.parametrize('world,mode', [(2, 'bucket'), (2, 'flat'), (4, 'bucket'), (2, 'one_range')])
def test_n_rank_equals_one_rank_on_concatenated_batch(tmp_path, world, mode):
    """bucket = the four gradient ranges all-reduced as the backward pass reports them (losses riding in the
    tail of the first one); flat = one all-reduce of the whole buffer; one_range = the small-batch frame path, where the
    library reports the whole buffer as ONE range after its single weight-gradient launch (the worker counts the
    collectives of every step: 4 / 1 / 1).  Shards carry different content."""
    F, steps = 8, 3
    outn, out1 = str(tmp_path / 'pn.npy'), str(tmp_path / 'p1.npy')
    launch(world, F, steps, outn, mode)
    launch(1, F, steps, out1, mode)
    got, want = np.load(outn), np.load(out1)
    p0 = O.flatten_params(O.init_params(SMALL_ARCH, 10)).astype(np.float64)
    d_got, d_want = got[:-3] - p0, want[:-3] - p0
    assert np.abs(d_want).max() > 0
    assert np.abs(d_got - d_want).max() / np.abs(d_want).max() < 1e-5
    assert np.allclose(got[-3:], want[-3:], rtol=1e-9)      # mean over ranks of the shard means == global mean


def test_mean_losses_equal_global_batch_losses(tmp_path):
    """The mean over ranks of the per-shard mean losses == the losses of the concatenated batch."""
    F, steps = 8, 2
    out2, out4 = str(tmp_path / 'p2.npy'), str(tmp_path / 'p4.npy')
    launch(2, F, steps, out2)
    launch(4, F, steps, out4)
    a, b = np.load(out2), np.load(out4)
    assert np.allclose(a[-3:], b[-3:], rtol=1e-9)
    assert np.abs(a[:-3] - b[:-3]).max() < 1e-9


def test_ranks_draw_independent_sampler_noise(tmp_path):
    """Seeded sampler: rank r uses key seed*world + r and step t uses counter t, so no two (rank, step) pairs share
    a draw (the reference draws independent N(0,1) per frame, util/layers.py:154)."""
    out = str(tmp_path / 'p.npy')
    launch(2, 8, 3, out, 'seeded')
    d0, d1 = np.load(out + '.draw0.npy'), np.load(out + '.draw1.npy')
    assert d0.shape == d1.shape == (3, 4, SMALL_ARCH['z_dim'])
    allv = np.concatenate([d0, d1]).reshape(6, -1)
    c = np.corrcoef(allv)
    assert np.abs(c - np.eye(6)).max() < 0.6 and len({v.tobytes() for v in allv}) == 6


@pytest.mark.parametrize('world', [1, 2])
def test_trainer_loop_status_is_collective_free_and_restore_continues(tmp_path, world):
    """VAETrainer.train with rank-dependent status intervals (a collective in the status path would pair with
    another rank's gradient all-reduce), training.log line format, checkpoint name, restore + continue."""
    launch(world, 8, 3, str(tmp_path / 't.npy'), 'trainer')


def test_callback_failure_surfaces_and_nothing_is_applied(tmp_path):
    """ctypes swallows exceptions raised in the bucket callback: the stepper must re-raise them after the library call,
    leave parameters and step counter untouched, and refuse reported ranges that do not tile the gradient buffer."""
    launch(2, 8, 1, str(tmp_path / 'c.npy'), 'cb_error')
