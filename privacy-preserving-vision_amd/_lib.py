"""ctypes binding of libppv_hip.so (include/ppv_hip.h).  The product path has NO fallback:
if the shared object is missing or a symbol is absent, importing a kernel wrapper raises."""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPV_LIB_PATH") or os.path.join(_HERE, "lib", "libppv_hip.so")   # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ppv_hip.h")

_c = ctypes
_P, _I, _L, _F, _Z = _c.c_void_p, _c.c_int, _c.c_long, _c.c_float, _c.c_size_t

ABI_VERSION = 21
PPV_ERR_NULL, PPV_ERR_BAD_SIZE, PPV_ERR_INIT, PPV_ERR_WORKSPACE = -1001, -1002, -1003, -1004   # include/ppv_hip.h

# name -> (restype, argtypes); mirrors include/ppv_hip.h (tests check the two agree)
PROTOTYPES = {
    "ppv_abi_version": (_I, []),
    "ppv_init": (_I, []),
    "ppv_fftconv_workspace_bytes": (_Z, [_I, _I, _I]),
    "ppv_otf_elems": (_Z, [_I, _I]),
    "ppv_otf_build": (_I, [_P, _I, _L, _L, _L, _I, _I, _I, _P, _P, _P]),
    "ppv_fftconv_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ppv_fftconv_fwd_u8": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ppv_fftconv_partials_per_image": (_I, [_I, _I, _I]),
    "ppv_group_max": (_I, [_P, _P, _I, _I, _P]),
    "ppv_div_by_group": (_I, [_P, _P, _L, _I, _P]),
    "ppv_fftconv_bwd_workspace_bytes": (_Z, [_I, _I, _I]),
    "ppv_fftconv_ic_partials": (_I, [_I, _I, _I]),
    "ppv_fftconv_ic_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "ppv_fftconv_ic_fwd_p": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_fftconv_ic_bwd_workspace_bytes_p": (_Z, [_I, _I, _I, _I]),
    "ppv_fftconv_ic_bwd_p": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _L, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_sensor_dot_count": (_I, [_P, _P, _P, _L, _P]),
    "ppv_fftconv_ic_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _L, _P, _P, _I, _I, _I, _P]),
    "ppv_fftconv_ic_bwd_u8": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _L, _P, _P, _I, _I, _I, _P]),
    "ppv_ic_psf_state_bytes": (_Z, [_I, _I, _I]),
    "ppv_ic_psf_fwd": (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "ppv_ic_psf_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "ppv_ic_psf_state_init": (_I, [_P, _I, _I, _I, _P]),
    "ppv_ic_psf_mark_support": (_I, [_P, _P, _I, _I, _I, _P]),
    "ppv_ic_psf_symmetric": (_I, [_P, _I, _I, _I, _P]),
    "ppv_ic_psf_state_offsets": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "ppv_ic_psf_fields_f32": (_I, []),
    "ppv_ic_psf_set_fields_f32": (_I, [_I]),
    "ppv_zernike_basis": (_I, [_P, _P, _P, _I, _I, _c.c_double, _c.c_double, _P]),
    "ppv_zernike_max_order": (_I, []),
    "ppv_fftconv_fd_bwd_workspace_bytes": (_Z, [_I, _I, _I]),
    "ppv_fftconv_fd_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ppv_fd_psf_bwd_workspace_bytes": (_Z, [_I]),
    "ppv_fd_psf_bwd": (_I, [_P] * 10 + [_F, _F, _P, _P, _P, _P, _P, _I, _P]),
    "ppv_zernike_grad_scratch_bytes": (_Z, [_I, _L]),
    "ppv_zernike_grad": (_I, [_P, _P, _P, _P, _I, _L, _P]),
    "ppv_stem_conv6": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ppv_fan_input": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_bilinear_resize_fwd": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "ppv_bilinear_resize_bwd": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "ppv_avgpool2_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ppv_upsample2_add": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ppv_concat3_add": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ppv_bn_act_split3": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "ppv_im2col_split": (_I, [_P, _P] + [_I] * 12 + [_P]),
    "ppv_pad_split": (_I, [_P, _P, _L, _I, _I, _I, _P]),
    "ppv_split3_rows": (_I, [_P, _L, _P, _L, _I, _I, _I, _P]),
    "ppv_fan_head": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ppv_ssim_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_ssim_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_conv3x3_bnin_supported": (_I, [_I, _I, _I, _I, _I]),
    "ppv_conv3x3_bnin": (_I, [_P, _P, _I, _c.c_double, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "ppv_bn_f32_workspace_bytes": (_Z, [_I]),
    "ppv_split6_rows": (_I, [_P, _P, _L, _I, _I, _P]),
    "ppv_bn_f32_fwd": (_I, [_P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "ppv_bn_f32_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "ppv_maxpool_f32_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_maxpool_f32_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_adaptive_pool_f32_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ppv_adaptive_pool_f32_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ppv_mse_workspace_bytes": (_Z, []),
    "ppv_mse_fwd": (_I, [_P, _P, _L, _P, _P, _P]),
    "ppv_mse_bwd": (_I, [_P, _P, _P, _P, _F, _P, _P, _L, _P]),
    "ppv_dec_prepare": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ppv_dec_attend_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ppv_dec_attend_bwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_lstm_cell_fwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "ppv_lstm_cell_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "ppv_dec_combine": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ppv_decc_attend_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ppv_decc_attend_bwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ppv_dec_enc_grad": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_decc_enc_grad": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ppv_decc_mean": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ppv_corr_volume": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ppv_alt_corr_fwd": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "ppv_alt_corr_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P]),
    "ppv_avgpool2": (_I, [_P, _P, _L, _I, _I, _P]),
    "ppv_corr_lookup_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ppv_avgpool2_bwd_acc": (_I, [_P, _P, _L, _I, _I, _P]),
    "ppv_corr_volume_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ppv_corr_lookup_all": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_corr_lookup": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ppv_zernike_contract": (_I, [_P, _P, _P, _I, _L, _P]),
    "ppv_fd_psf_workspace_bytes": (_Z, [_I]),
    "ppv_fd_psf_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _P]),
    "ppv_conv_gemm": (_I, [_P, _P, _P, _P, _P, _P, _P] + [_I] * 14 + [_P]),
    "ppv_conv_gemm_red": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P] + [_I] * 13 + [_P]),
    "ppv_weight_layout_multi": (_I, [_P, _I, _I, _P]),
    "ppv_adam_multi": (_I, [_P, _I, _I] + [_c.c_double] * 7 + [_P]),
    "ppv_weight_layout": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ppv_conv_stat_tiles": (_I, [_L]),
    "ppv_conv_bn_relu_coop": (_I, [_P] * 10 + [_F, _F, _P, _P] + [_I] * 6 + [_P]),
    "ppv_instnorm_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "ppv_instnorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "ppv_gemm_f32_ksplit": (_I, [_I, _I, _I]),
    "ppv_gemm_f32": (_I, [_P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _I, _P]),
    "ppv_gemm_f32_ws_plan": (_I, [_I, _I, _I, _P]),
    "ppv_gemm_f32_tn_plan": (_I, [_I, _I, _I, _P]),
    "ppv_gemm_f32_tn": (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _P, _P]),
    "ppv_gemm_bf16x3_nt_plan": (_I, [_I, _I, _I, _P]),
    "ppv_gemm_bf16x3_nt": (_I, [_P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _I, _P, _P]),
    "ppv_gemm_bf16x3_tn_plan": (_I, [_I, _I, _I, _P]),
    "ppv_gemm_bf16x3_tn": (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _P, _P]),
    "ppv_gemm_f32_ws": (_I, [_P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _I, _P, _P]),
    "ppv_conv_gemm_rect": (_I, [_P, _P, _P, _P] + [_I] * 10 + [_P]),
    "ppv_gru_zr": (_I, [_P, _I, _P, _P, _P, _P, _L, _I, _P]),
    "ppv_gru_out": (_I, [_P, _I, _P, _P, _P, _P, _L, _I, _P]),
    "ppv_conv_set_variant": (_I, [_I]),
    "ppv_conv_wgrad_scratch_bytes": (_Z, [_L, _I, _I, _I, _I]),
    "ppv_conv_wgrad_pair_supported": (_I, [_I] * 9),
    "ppv_conv_wgrad_pair": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P, _I, _P]),
    "ppv_conv_wgrad": (_I, [_P, _P, _P, _P, _P] + [_I] * 11 + [_P]),
    "ppv_conv_wgrad_group": (_I, [_P, _P, _P, _I, _P] + [_I] * 5 + [_P]),
    "ppv_wgrad_set_variant": (_I, [_I]),
    "ppv_stem_weight_layout": (_I, [_P, _P, _I, _P]),
    "ppv_stem_conv": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_stem_dgrad_scatter": (_I, [_P, _P, _I, _I, _I, _P]),
    "ppv_stem_dgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ppv_bn_finalize": (_I, [_P, _I, _c.c_double, _P, _P, _P, _P, _F, _F, _P, _I, _P]),
    "ppv_bn_act": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _L, _P]),
    "ppv_bn_act_fold": (_I, [_P, _P, _c.c_double, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "ppv_bn_act_fold_rows": (_I, [_P, _P, _I, _c.c_double, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "ppv_bn_act_train": (_I, [_P, _P, _I, _c.c_double, _P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _L, _I, _I, _I, _P]),
    "ppv_bn_bwd_blocks": (_I, [_L, _I]),
    "ppv_bn_bwd": (_I, [_P, _P, _P, _P, _c.c_double, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "ppv_bn_bwd_sums2": (_I, [_P, _P, _P, _c.c_double, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P]),
    "ppv_bn_relu_maxpool": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_maxpool_relu_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_maxpool_bn_bwd": (_I, [_P, _P, _P, _P, _P, _c.c_double, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ppv_adaptive_pool_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ppv_adaptive_pool_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
}

class BottleneckFwd(ctypes.Structure):
    """include/ppv_hip.h PpvBottleneckFwd (field order = the header's)."""
    _fields_ = ([(n, _P) for n in ("xin", "w1", "w2", "w3", "x1", "y1", "x2", "y2", "x3", "yout", "bits", "stats1", "stats2", "stats3",
                                   "coef1", "coef2", "coef3", "g1", "b1", "rm1", "rv1", "g2", "b2", "rm2", "rv2", "g3", "b3", "rm3", "rv3",
                                   "zero_page")]
                + [(n, _F) for n in ("mom1", "eps1", "mom2", "eps2", "mom3", "eps3")]
                + [(n, _I) for n in ("B", "H", "W", "Cin", "planes", "stride", "T1", "T2", "T3")])


PROTOTYPES["ppv_bottleneck_fwd"] = (_I, [ctypes.POINTER(BottleneckFwd), _P])


class BottleneckBwd(ctypes.Structure):
    """include/ppv_hip.h PpvBottleneckBwd."""
    _fields_ = ([(n, _P) for n in ("g", "xin", "x1", "y1", "x2", "y2", "x3", "xin_bits", "c1", "c2", "c3", "wd1", "wd2", "wd3",
                                   "part3", "part2", "part1", "kc3", "kc2", "kc1", "gx3", "gy2", "gx2", "gy1", "gx1", "gin",
                                   "dg3", "db3", "dg2", "db2", "dg1", "db1", "dw3", "dw2", "dw1", "wscratch", "x3_prev", "part3_prev",
                                   "zero_page")]
                + [(n, _I) for n in ("B", "H", "W", "planes", "part3_ready", "red2", "red1")]
                + [("_pad", _I), ("wstride", _L)])


PROTOTYPES["ppv_bottleneck_bwd"] = (_I, [ctypes.POINTER(BottleneckBwd), _P, _P])


class WgradReduce(ctypes.Structure):
    """include/ppv_hip.h PpvWgradReduce."""
    _fields_ = [("slabs", _P), ("out", _P)] + [(n, _I) for n in ("N", "C", "R", "S", "nslab", "TN", "mode", "blocks")]


PROTOTYPES["ppv_conv_wgrad_ex"] = (_I, [_P, _P, _P, _P, _P] + [_I] * 11 + [_P, ctypes.POINTER(WgradReduce)])
PROTOTYPES["ppv_wgrad_reduce_multi"] = (_I, [ctypes.POINTER(WgradReduce), _I, _P])

TRUNK_MAX_BLOCKS = 64


class TrunkBlock(ctypes.Structure):
    """include/ppv_hip.h PpvTrunkBlock."""
    _fields_ = [(n, _I) for n in ("planes", "stride", "proj", "train_w")]


class TrunkDesc(ctypes.Structure):
    """include/ppv_hip.h PpvTrunkDesc."""
    _fields_ = [(n, _I) for n in ("B", "H", "W", "nblocks", "fold_rows", "wgrad_reduce3", "_r0", "_r1")] + [("blk", TrunkBlock * TRUNK_MAX_BLOCKS)]


TRUNK_CONV_FIELDS = ("wt", "wd", "gamma", "beta", "rm", "rv", "dw", "dgamma", "dbeta")    # PpvTrunkConv: nine pointers per convolution
# the conv / hyper tables cross the boundary as flat arrays (c_uint64 * (9 n), c_float * (2 n)): filled with slice assignments
PROTOTYPES["ppv_trunk_arena_bytes"] = (_Z, [ctypes.POINTER(TrunkDesc)])
PROTOTYPES["ppv_trunk_block_offsets"] = (_I, [ctypes.POINTER(TrunkDesc), _I, _P])
PROTOTYPES["ppv_trunk_fwd"] = (_I, [ctypes.POINTER(TrunkDesc), _P, _P, _P, _P, _P, _P, _P])
PROTOTYPES["ppv_trunk_bwd"] = (_I, [ctypes.POINTER(TrunkDesc), _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P])
PROTOTYPES["ppv_stream_fork"] = (_I, [_P, _P])
PROTOTYPES["ppv_stream_create_masked"] = (_I, [ctypes.POINTER(_P), _I, _I])
PROTOTYPES["ppv_stream_destroy"] = (_I, [_P])

_lib = None


def header_symbols():
    """Function names declared in include/ppv_hip.h."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ppv_[a-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if L.ppv_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH}: ABI version {L.ppv_abi_version()} != {ABI_VERSION} expected by this package "
                               "(stale build? re-run __graft_entry__.build())")
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f"libppv_hip: {what} failed with status {status}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The current HIP stream of the current device as a void*.  torch.cuda.current_stream() costs ~8 us per call (device-index
    resolution, a Stream object): at ~1600 kernel launches per step that was 5 ms of host time; the raw accessors are ~0.3 us."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libppv_hip kernels need device tensors (no CPU path in the product)")
