import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="session", autouse=True)
def _release_scalar_env_pool():
    """qg_env_destroy parks handles in a process-wide pool for the next clone; the suite hands them back at the end (qg_env_pool_clear)."""
    yield
    try:
        from qiskit_gym_amd import _lib

        if _lib._lib is not None:
            _lib._lib.qg_env_pool_clear()
    except Exception:
        pass
