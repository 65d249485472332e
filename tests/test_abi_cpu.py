"""CPU: the C-ABI shared object loads and exports every symbol include/ppv_hip.h declares (no compute)."""
import os

import pytest


def test_library_exports_header_symbols():
    import ppv_amd
    from ppv_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.lib()
    names = _lib.header_symbols()
    assert len(names) >= 8
    for n in names:
        assert hasattr(L, n), n
        assert n in _lib.PROTOTYPES, f"{n} declared in the header but not bound in _lib.PROTOTYPES"
    for n in _lib.PROTOTYPES:
        assert n in names, f"{n} bound in Python but missing from include/ppv_hip.h"
    assert L.ppv_abi_version() >= 1
    assert L.ppv_otf_elems(3, 512) == 3 * 257 * 512


def test_product_has_no_cpu_fallback():
    import torch
    import ppv_amd.fftconv as fc
    with pytest.raises(RuntimeError):
        fc.otf_build(torch.zeros(3, 256, 256), 256, 512)
