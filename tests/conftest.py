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
def one_adder_stats(monkeypatch):
    """Train-mode forward passes that must agree BIT FOR BIT across two runs: give the convolutions as many partial rows for their
    BatchNorm statistics as they have row tiles (one f32 adder per address).  With the default two rows (encoder.py PPV_BN_FOLD_ROWS)
    the order of the atomics is open and two passes may differ by an ulp of a statistic."""
    monkeypatch.setenv("PPV_BN_FOLD_ROWS", "32")


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


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu()
        if x.is_complex():
            import torch
            x = torch.view_as_real(x.resolve_conj())
        x = x.numpy()
    return np.asarray(x, dtype=np.float64)


def rel_err(a, b):
    a = _np(a)
    b = _np(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
