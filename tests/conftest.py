"""pytest config: registers the `gpu` marker and puts the product package + repo root on sys.path.

`-m "not gpu"` : oracle vs the reference's golden vectors, host logic, C-ABI symbol check, gloo world_size-2 tests.
`-m gpu`       : parity tests proper -- HIP path (through the C-ABI) vs oracle / goldens on a real MI355X.
Nothing here (or in any test) reads /root/reference: it does not exist on the GPU box.
"""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'physics-aware-multiplex-gnn_amd')
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


def maxnorm_err(a, b):
    """max|a-b| / max|b|  -- the parity metric."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
