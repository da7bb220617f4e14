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


@pytest.fixture
def chunked_kernel(monkeypatch):
    """Small problems (< 6144 frames) normally go to the one-frame-per-wave kernel; tests of the fused, chunk-walking
    kernel on small shapes switch that rule off (SPECINV_SMALL_FRAMES=0) and keep the plan cache from mixing the two."""
    from spectrogram_inversion_amd.plan import clear_plan_cache
    monkeypatch.setenv("SPECINV_SMALL_FRAMES", "0")
    clear_plan_cache()
    yield
    clear_plan_cache()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def hann(n, dtype=np.float32):
    return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(dtype)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def sc_linear(db):
    return 10.0 ** (np.asarray(db, dtype=np.float64) / 20.0)
