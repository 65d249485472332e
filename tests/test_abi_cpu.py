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


def test_entry_points_reject_bad_arguments_before_any_launch():
    """Error behaviour of the C ABI: null pointers and unsupported sizes come back as PPV_ERR_* codes from the argument checks,
    before any HIP call (so this runs without a GPU)."""
    import ctypes
    from ppv_amd import _lib
    L = _lib.lib()
    ERR_NULL, ERR_SIZE = _lib.PPV_ERR_NULL, _lib.PPV_ERR_BAD_SIZE
    p = ctypes.c_void_p(64)          # a non-null pointer that is never dereferenced on these paths
    # conv: nulls; channel counts off the 64 grid; statistics without rows; fused BN sums need 64-column-aligned N and a buffer
    assert L.ppv_conv_gemm(None, p, p, None, None, None, p, 1, 8, 8, 64, 8, 8, 64, 1, 1, 1, 0, 1, 0, 0, None) == ERR_NULL
    assert L.ppv_conv_gemm(p, p, p, None, None, None, p, 1, 8, 8, 48, 8, 8, 64, 1, 1, 1, 0, 1, 0, 0, None) == ERR_SIZE
    assert L.ppv_conv_gemm(p, p, p, p, None, None, p, 1, 8, 8, 64, 8, 8, 64, 1, 1, 1, 0, 1, 0, 0, None) == ERR_SIZE
    assert L.ppv_conv_gemm(p, p, p, p, p, None, p, 1, 8, 8, 64, 8, 8, 64, 1, 1, 1, 0, 1, 0, 4, None) == ERR_SIZE
    assert L.ppv_conv_gemm_red(p, p, p, None, p, None, None, None, p, 1, 8, 8, 64, 8, 8, 128, 1, 1, 1, 0, 1, 8, None) == ERR_NULL
    assert L.ppv_conv_gemm_red(p, p, p, p, p, None, None, None, p, 1, 8, 8, 64, 8, 8, 96, 1, 1, 1, 0, 1, 8, None) == ERR_SIZE
    assert L.ppv_conv_gemm_red(p, p, p, p, p, p, p, None, p, 1, 8, 8, 64, 8, 8, 128, 1, 1, 1, 0, 1, 8, None) == ERR_SIZE
    # BN backward: channel grid; pre-reduced sums only with relu-free... (size check) and nulls
    assert L.ppv_bn_bwd(None, None, p, p, 1.0, p, None, None, None, p, p, 8, 64, 0, 0, None) == ERR_NULL
    assert L.ppv_bn_bwd(p, None, p, p, 1.0, p, None, None, None, p, p, 8, 96, 0, 0, None) == ERR_SIZE
    assert L.ppv_bn_bwd(p, None, p, p, 1.0, p, None, None, None, p, p, 8, 64, 1, 0, None) == ERR_NULL      # relu = 1 needs y
    # stem: odd sizes, widths the row-staged data gradient does not cover
    assert L.ppv_stem_conv(p, p, p, None, 0, 1, 15, 16, None) == ERR_SIZE
    assert L.ppv_stem_dgrad(p, p, p, p, 1, 8, 64, None) == ERR_SIZE
    assert L.ppv_stem_dgrad(p, p, None, p, 1, 8, 128, None) == ERR_NULL
