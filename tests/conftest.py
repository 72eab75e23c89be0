import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# The oracle's OpenMP loops default to every CPU the machine shows (256 on the GPU box, of which the container may run 16): cap the team at
# what this process may actually use.  Set here, before any test module imports torch or the oracle: libgomp reads it once, when it is loaded.
if "OMP_NUM_THREADS" not in os.environ:
    try:
        _q, _p = open("/sys/fs/cgroup/cpu.max").read().split()
        _quota = None if _q == "max" else int(float(_q) / float(_p) + 0.5)
    except Exception:
        _quota = None
    _n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(_n, _quota or _n)))


# tools/run_cpu_tests_asan.sh: the CPU tests against the sanitizer build of libqgym's host side (qiskit_gym_amd/lib/libqgym_asan.so)
if os.environ.get("QGYM_LIB_ASAN"):
    from qiskit_gym_amd import _lib as _qg_lib

    _qg_lib.LIB_PATH = os.path.join(ROOT, "qiskit_gym_amd", "lib", "libqgym_asan.so")


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

