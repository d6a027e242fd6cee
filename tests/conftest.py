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
    # A fresh checkout has no built artefacts (they are git-ignored): build the HIP library
    # (hipcc cross-compiles gfx950 without a GPU) and the CPU oracle before collection, exactly
    # what __graft_entry__.build() does.
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "spectavi_amd", "libspectavi.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "spectavi_amd", "csrc"), "-j", "4", "-s"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])


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
