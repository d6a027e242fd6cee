"""GPU: a short slice of the randomised differential run (tests/fuzz_gpu.py) inside the suite:
a few seconds of random shapes / parameters / value ranges per entry point, every case compared
with the CPU oracle bit for bit.  A mismatch aborts with the reproducing (seed, path, case)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(os.path.dirname(__file__), "fuzz_gpu.py"))
fuzz = importlib.util.module_from_spec(_spec)


@pytest.fixture(scope="module")
def fz(oracle):
    _spec.loader.exec_module(fuzz)
    return fuzz


@pytest.mark.parametrize("path,seconds", [("l1k2", 4), ("cascade", 6), ("dlt", 3), ("ratio", 2), ("score", 2),
                                          ("normalize", 3)])
def test_random_cases_match_oracle(fz, path, seconds):
    fn = getattr(fz, "fuzz_" + path)
    try:
        cases = fn(20261004, float(seconds))
    except SystemExit as e:  # the tool reports a mismatch by exiting with the failing configuration
        pytest.fail(str(e))
    assert cases >= 1
