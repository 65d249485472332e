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


def _resnet101_desc(B=128, H=256, W=256):
    import ctypes
    from ppv_amd import _lib
    d = _lib.TrunkDesc()
    d.B, d.H, d.W, d.fold_rows = B, H, W, 2
    blocks = []
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 23, 3))):
        for b in range(n):
            blocks.append((planes, 2 if (b == 0 and li > 0) else 1, int(b == 0), 0 if li == 0 else 15))
    d.nblocks = len(blocks)
    for i, (p, s, pr, tw) in enumerate(blocks):
        k = d.blk[i]
        k.planes, k.stride, k.proj, k.train_w = p, s, pr, tw
    return d, ctypes.byref(d)


def test_trunk_executor_arena_layout_is_a_pure_function_of_the_descriptor():
    """ppv_trunk_arena_bytes / ppv_trunk_block_offsets (csrc/trunk_plan.hip) run on the host only: ResNet-101 at the benchmark geometry needs
    one arena of ~20 GB; every tensor of every block starts on a 256-byte boundary, inside the arena, and no two of them overlap; bad
    geometries are refused with a status code (arena size 0), nothing is launched."""
    import ctypes
    from ppv_amd import _lib
    L = _lib.lib()
    d, ref = _resnet101_desc()
    total = L.ppv_trunk_arena_bytes(ref)
    assert 15e9 < total < 25e9
    d4, ref4 = _resnet101_desc(B=4)
    assert 0 < L.ppv_trunk_arena_bytes(ref4) < total / 16
    out = (ctypes.c_size_t * 20)()
    spans = []
    M = {0: 128 * 64 * 64}
    sizes_seen = 0
    for blk in range(-1, d.nblocks):
        assert L.ppv_trunk_block_offsets(ref, blk, out) == 0
        offs = [int(v) for v in out]
        for o in offs:
            assert o % 256 == 0 and o < total
        spans += [o for o in offs if o]
        sizes_seen += 1
    assert len(spans) == len(set(spans)), "two tensors of the arena share an offset"
    assert L.ppv_trunk_block_offsets(ref, d.nblocks, out) == _lib.PPV_ERR_BAD_SIZE
    # refused geometries: image not a multiple of 32, identity block whose input width is not 4 x planes, zero blocks
    bad, rb = _resnet101_desc(H=250)
    assert L.ppv_trunk_arena_bytes(rb) == 0
    bad, rb = _resnet101_desc()
    bad.blk[1].planes = 128
    assert L.ppv_trunk_arena_bytes(rb) == 0
    bad, rb = _resnet101_desc()
    bad.nblocks = 0
    assert L.ppv_trunk_arena_bytes(rb) == 0
    # entry points check their pointers before touching the device
    p = ctypes.c_void_p(64)
    assert L.ppv_trunk_fwd(ref, None, p, p, p, p, p, None) == _lib.PPV_ERR_NULL
    assert L.ppv_trunk_bwd(ref, p, None, p, p, 1, 36, None, p, 0, 33, None, None) == _lib.PPV_ERR_NULL
    assert L.ppv_stream_create_masked(None, 0, 64) == _lib.PPV_ERR_NULL
    h = ctypes.c_void_p()
    assert L.ppv_stream_create_masked(ctypes.byref(h), 200, 100) == _lib.PPV_ERR_BAD_SIZE
