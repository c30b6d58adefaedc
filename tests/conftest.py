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


@pytest.fixture(scope='session')
def arch():
    with open(os.path.join(PKG, 'architecture-vae-vcc2016.json')) as fp:
        return json.load(fp)


@pytest.fixture(scope='session')
def small_arch():
    from helpers import SMALL_ARCH
    return SMALL_ARCH
