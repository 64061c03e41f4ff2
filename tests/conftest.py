import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """CPU oracle, transcendental functions through libm like the Rust reference."""
    from oracle_loader import oracle_backend
    return oracle_backend(det=False)


@pytest.fixture(scope="session")
def orc_det():
    """CPU oracle built with the product's deterministic math (bit-exact GPU comparisons)."""
    from oracle_loader import oracle_backend
    return oracle_backend(det=True)


@pytest.fixture(scope="session")
def ftn():
    """The HIP product library. Loading needs no GPU; compute calls do."""
    from fountain_amd import default_backend
    return default_backend()


@pytest.fixture(scope="session")
def gpu(ftn):
    n = ftn.fn("device_count")()
    if n < 1:
        pytest.fail("GPU test selected but no HIP device is visible: the HIP path has no CPU fallback")
    return ftn
