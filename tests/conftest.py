import json
import os
import sys

import pytest
import torch

torch.set_num_threads(min(8, os.cpu_count() or 1))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'vae-npvc_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


# Order of the GPU gate (the driver runs `pytest -x -q -m gpu`): the float64-oracle parity of the kernels the bench times
# first, then the small-batch frame kernels, the plugin / runtime plumbing, and the parity-unpinned VAWGAN branch last,
# so that a failure in a "next" row can never hide the hot path's tests behind -x.
_FILE_ORDER = ['test_gpu_parity.py', 'test_gpu_frame.py', 'test_gpu_fgroup.py', 'test_gpu_runtime.py', 'test_gpu_plugins.py',
               'test_gpu_vawgan.py']


def pytest_collection_modifyitems(session, config, items):
    def key(it):
        name = os.path.basename(str(it.fspath))
        return _FILE_ORDER.index(name) if name in _FILE_ORDER else -1      # CPU files keep their place in front
    items.sort(key=key)            # stable: the order inside a file is kept


@pytest.fixture(scope='session')
def arch():
    with open(os.path.join(PKG, 'architecture-vae-vcc2016.json')) as fp:
        return json.load(fp)


@pytest.fixture(scope='session')
def small_arch():
    from helpers import SMALL_ARCH
    return SMALL_ARCH
