import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: GiB-scale streams (still part of -m gpu)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def hip():
    """The product library through its ctypes binding; GPU tests fail (not skip) if it cannot run."""
    import aesgcm_amd  # noqa: F401
    from aesgcm_amd import lib
    lib.load()
    n = lib.device_count()          # raises AesGcmError(EHIP) without a device: fail loudly
    assert n >= 1
    return lib


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle
