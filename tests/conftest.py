import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Fixtures generated from the Python reference by tests/golden/make_golden.py."""
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_cpu.npz")))


def flatten_rows(x):
    """[R, S, ...] -> flat samples plus ray_indices / packed_info for the flattened layout."""
    x = np.asarray(x)
    R, S = x.shape[:2]
    ray_indices = np.repeat(np.arange(R, dtype=np.int64), S)
    packed_info = np.stack([np.arange(R, dtype=np.int64) * S, np.full(R, S, np.int64)], -1)
    return x.reshape((R * S,) + x.shape[2:]), ray_indices, packed_info
