"""world_size-2 data-parallel host logic under gloo on CPU.

The HIP engine needs a GPU, so the Stepper (hipvae/dp.py) is driven with a CPU stand-in
backend built from the ORACLE (tests/dp_gloo_worker.py): the thing under test is the
sharding / all-reduce / grad_scale / broadcast logic, not the arithmetic.
Property: N-rank parameters after k steps == 1-rank parameters on the
concatenated batch.
"""
import os
import subprocess
import sys

import numpy as np

from helpers import SMALL_ARCH
from oracle import convvae_oracle as O

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dp_gloo_worker.py')


def launch(world, F, steps, out, port):
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), str(F), str(steps), out])
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=180) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def test_two_rank_equals_one_rank_on_concatenated_batch(tmp_path):
    F, steps = 8, 3
    port = 29000 + os.getpid() % 2000
    out2, out1 = str(tmp_path / 'p2.npy'), str(tmp_path / 'p1.npy')
    launch(2, F, steps, out2, port)
    launch(1, F, steps, out1, port)
    got, want = np.load(out2), np.load(out1)
    p0 = O.flatten_params(O.init_params(SMALL_ARCH, 10)).astype(np.float64)
    d_got, d_want = got[:-3] - p0, want[:-3] - p0
    assert np.abs(d_want).max() > 0
    assert np.abs(d_got - d_want).max() / np.abs(d_want).max() < 1e-5
    assert np.allclose(got[-3:], want[-3:], rtol=1e-9)
