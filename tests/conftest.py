import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; builds oracle/liboracle.so with gcc if absent)."""
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def uniform_u8(seed, rows, dim):
    """The reference's test distribution (test/test_feature.py:112-115): uniform uint8."""
    return np.random.default_rng(seed).integers(0, 256, (rows, dim), dtype=np.uint8)
