import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through libbpmsm.so's C ABI)")


@pytest.fixture(scope="session")
def oracle_c():
    """Make sure the C part of the oracle is built (gcc, seconds)."""
    import subprocess
    from oracle import ec
    if not ec.have_c():
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    assert ec.have_c()
    return ec


@pytest.fixture(scope="session")
def ctx():
    """One bp_ctx on device 0 for the GPU tests (fails loudly if the library/GPU is missing)."""
    import bulletproofs_amd as bpa
    c = bpa.Context(0)
    yield c
    c.close()
