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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def volume_896():
    """Zernike volume [350,896,896] f32 (oracle generator), cached on disk for the session."""
    import tempfile
    from oracle.zernike import zernike_volume
    path = os.path.join(tempfile.gettempdir(), "ppv_zernike_volume_896_n350.npy")
    if os.path.exists(path):
        return np.load(path, mmap_mode="r")
    v = zernike_volume(896, 350).astype(np.float32)
    np.save(path, v)
    return v


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
