"""CPU stand-in engine for bench.py's launcher test (test infrastructure; built from the oracle).

bench.py --standin <this file> runs its self-launch / process-group / timing logic under gloo with this engine
instead of hipvae.Engine: nothing here is a measurement."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)
from helpers import SMALL_ARCH  # noqa: E402
from dp_gloo_worker import OracleBackend  # noqa: E402

ARCH = SMALL_ARCH


def make_engine(arch, args):
    rank = int(os.environ.get('RANK', '0'))
    return OracleBackend(arch, seed=10 + rank)      # different init per rank: the broadcast must fix it
