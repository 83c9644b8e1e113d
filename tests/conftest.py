import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box via gpurun)")


@pytest.fixture(scope="session")
def oracle_backend():
    import oracle

    return oracle.backend()


@pytest.fixture()
def use_oracle(oracle_backend):
    """Route the package's ops through the CPU oracle for the duration of one test (host-logic tests)."""
    from pointcloudpdf_amd import _native

    prev = _native._set_backend_for_testing(oracle_backend)
    yield oracle_backend
    _native._set_backend_for_testing(prev)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
