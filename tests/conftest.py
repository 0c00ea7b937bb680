import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through libbpmsm.so's C ABI)")
    # torch before libbpmsm.so: the wheel brings its own HIP runtime, and a process that loaded the system one first (through
    # the library) made torch.cuda report "No HIP GPUs are available" in the tests that move buffers with torch.  With every
    # test file collected test_dist_gloo's import did this by accident; a subset of files (pytest tests/test_gpu_verify.py
    # tests/test_zz_no_alloc.py) did not.  bench.py imports torch first as well.
    try:
        import torch  # noqa: F401
    except Exception:
        pass


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
