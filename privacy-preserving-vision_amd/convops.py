"""torch-facing wrappers of the trunk kernels (csrc/conv_gemm.hip, conv_wgrad_stem.hip, trunk_ops.hip).
Activations are NHWC bfloat16; statistics, BN coefficients and weight gradients are float32."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

_zero_pages = {}
BF16, F32 = torch.bfloat16, torch.float32


def zero_page(device):
    key = (device.type, device.index)
    if key not in _zero_pages:
        _apply_env_variants()
        _zero_pages[key] = torch.zeros(256, dtype=torch.uint8, device=device)
    return _zero_pages[key]


def L():
    return _lib.lib()


def _apply_env_variants():
    """tuning hooks (A/B runs): PPV_CONV_VARIANT / PPV_WGRAD_VARIANT, see csrc/conv_gemm.hip, conv_wgrad_stem.hip"""
    import os
    if "PPV_CONV_VARIANT" in os.environ:
        L().ppv_conv_set_variant(int(os.environ["PPV_CONV_VARIANT"], 0))
    if "PPV_WGRAD_VARIANT" in os.environ:
        L().ppv_wgrad_set_variant(int(os.environ["PPV_WGRAD_VARIANT"], 0))


# bench.py sets PROFILE = [] to collect (kernel, algorithmic flops, start event, end event) per conv launch
PROFILE = None


def _timed(kind, flops, launch, nbytes=0.0):
    """nbytes: the launch's compulsory HBM bytes (every operand and result once) for the roofline's HBM view."""
    if PROFILE is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = launch()
    e1.record()
    PROFILE.append((kind, flops, e0, e1, nbytes))
    return r


# ----------------------------------------------------------------------------- convolution
def weight_layout(w, mode):
    """w [Cout,Cin,R,S] f32 -> bf16 GEMM rows: mode 0 forward [Cout,R,S,Cin]; mode 1 dgrad [Cin,R,S,Cout] flipped."""
    Cout, Cin, R, S = w.shape
    shape = (Cout, R, S, Cin) if mode == 0 else (Cin, R, S, Cout)
    out = torch.empty(shape, dtype=BF16, device=w.device)
    check(L().ppv_weight_layout(ptr(w.contiguous()), ptr(out), Cout, Cin, R, S, mode, stream_ptr()), "ppv_weight_layout")
    return out


class WeightLayouts:
    """bf16 kernel layouts (forward + data-gradient) of a list of conv weights, refreshed by ONE launch."""

    def __init__(self, weights):
        import struct
        self.weights = list(weights)
        dev = self.weights[0].device
        self.fwd, self.dg = [], []
        recs, blk = [], 0
        for w in self.weights:
            Cout, Cin, R, S = w.shape
            f = torch.empty((Cout, R, S, Cin), dtype=BF16, device=dev)
            d = torch.empty((Cin, R, S, Cout), dtype=BF16, device=dev)
            self.fwd.append(f)
            self.dg.append(d)
            recs.append(struct.pack("<QQQiiiii", w.data_ptr(), f.data_ptr(), d.data_ptr(), Cout, Cin, R, S, blk))
            recs[-1] += b"\0" * (48 - len(recs[-1]))
            if Cout % 32 or Cin % 32 or R * S > 9:
                raise ValueError("WeightLayouts: channel counts must be multiples of 32 and the kernel at most 3x3")
            blk += (Cout // 32) * (Cin // 32)
        self.total_blocks = blk
        import numpy as np
        self.desc = torch.from_numpy(np.frombuffer(b"".join(recs), dtype=np.uint8).copy()).to(dev)
        self._ptrs = [w.data_ptr() for w in self.weights]

    def valid(self):
        return all(w.data_ptr() == p for w, p in zip(self.weights, self._ptrs))

    def refresh(self):
        check(L().ppv_weight_layout_multi(ptr(self.desc), len(self.weights), self.total_blocks, stream_ptr()),
              "ppv_weight_layout_multi")


def stat_tiles(M):
    return L().ppv_conv_stat_tiles(M)


def conv_fwd(x, wt, stride, pad, stat_part=None, out_f32=False):
    """x [B,H,W,Cin] bf16, wt [Cout,R,S,Cin] bf16 -> [B,Ho,Wo,Cout]; stat_part [tiles,2,Cout] f32 BN partials."""
    B, H, W, Cin = x.shape
    Cout, R, S, _ = wt.shape
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), dtype=F32 if out_f32 else BF16, device=x.device)
    kind = "conv_gemm<128>" if Cout % 128 == 0 else "conv_gemm<64>"
    _timed(kind, 2.0 * B * Ho * Wo * Cout * R * S * Cin, lambda: check(
        L().ppv_conv_gemm(ptr(x), ptr(wt), ptr(out), ptr(stat_part), None, None, ptr(zero_page(x.device)), B, H, W, Cin,
                          Ho, Wo, Cout, R, S, stride, -pad, 1, int(out_f32),
                          0 if stat_part is None else stat_part.shape[0], stream_ptr()), "ppv_conv_gemm"),
           nbytes=x.numel() * 2.0 + wt.numel() * 2.0 + out.numel() * out.element_size())
    return out


RED_ROWS = 8       # partial rows of the BN-backward sums (ppv_conv_gemm_red / ppv_bn_bwd's fused apply)


def red_supported(rows, C):
    """Whether conv_dgrad(..., red=...) exists for a gradient of `rows` pixels x C channels (the tiles that carry the sums)."""
    return C % 128 == 0 or (C % 64 == 0 and rows >= 128 * 1024)


def _stream_kernel(K, N, R, S, div, addend, out_f32):
    """Mirror of ppv_conv_gemm's automatic rule (csrc/conv_gemm.hip / conv_stream.hip): which launches the streaming 1x1 kernel
    takes -- only used to label launches for bench.py's per-class roofline."""
    return R == 1 and S == 1 and div == 1 and K in (64, 128, 256) and N % 64 == 0 and N >= 2 * K and addend is not None and not out_f32


def conv_dgrad(g, wd, stride, pad, in_hw, addend=None, out_f32=False, relu_bits=None, red=None):
    """g [B,Ho,Wo,Cout] bf16, wd [Cin,R,S,Cout] bf16 (flipped) -> grad wrt the conv input [B,H,W,Cin] (+ addend);
    relu_bits: (conv input > 0) bit mask from bn_act(..., want_bits=True) when that input is a ReLU output -> lanes whose bit
    is clear get a zero gradient.  red = (x_raw, part[, coef]): also take the BN-backward sums of the stored gradient against
    x_raw (the raw conv output of the BatchNorm this gradient flows into) into the PRE-ZEROED f32 part [>= 16 * Cin];
    bn_bwd(..., part=part, part_ready=True) then skips its reduce pass.  With coef (that BatchNorm's bn_finalize output, BN +
    ReLU without residual) the ReLU mask is recomputed from x_raw and applied to the returned gradient: bn_bwd takes relu=0."""
    B, Ho, Wo, Cout = g.shape
    Cin, R, S, _ = wd.shape
    H, W = in_hw
    out = torch.empty((B, H, W, Cin), dtype=F32 if out_f32 else BF16, device=g.device)
    kind = "conv_gemm<128>" if Cin % 128 == 0 else "conv_gemm<64>"
    if _stream_kernel(Cout, Cin, R, S, stride, addend, out_f32):
        kind = "conv1x1_stream"
    if red is not None:
        xr, part = red[0], red[1]
        rcoef = red[2] if len(red) > 2 else None
        assert xr.shape == out.shape and xr.dtype == BF16 and not out_f32
        # its own class in the profile: these launches also do the BatchNorm-backward reduction (a different kernel instantiation)
        _timed(kind + "+bn_sums", 2.0 * B * Ho * Wo * Cout * R * S * Cin, lambda: check(
            L().ppv_conv_gemm_red(ptr(g), ptr(wd), ptr(out), ptr(part), ptr(xr), ptr(rcoef), ptr(addend), ptr(relu_bits),
                                  ptr(zero_page(g.device)), B, Ho, Wo, Cout, H, W, Cin, R, S, 1, -(R - 1 - pad), stride,
                                  RED_ROWS, stream_ptr()), "ppv_conv_gemm_red"),
               nbytes=(g.numel() + wd.numel() + 2 * out.numel() + (out.numel() if addend is not None else 0)) * 2.0
               + (out.numel() / 8 if relu_bits is not None else 0))
        return out
    # algorithmic flops of the data gradient = those of the forward conv it differentiates
    _timed(kind, 2.0 * B * Ho * Wo * Cout * R * S * Cin, lambda: check(
        L().ppv_conv_gemm(ptr(g), ptr(wd), ptr(out), None, ptr(addend), ptr(relu_bits), ptr(zero_page(g.device)), B, Ho, Wo, Cout,
                          H, W, Cin, R, S, 1, -(R - 1 - pad), stride, int(out_f32), 0, stream_ptr()), "ppv_conv_gemm"),
           nbytes=(g.numel() + wd.numel() + (out.numel() if addend is not None else 0)) * 2.0 + out.numel() * out.element_size()
           + (out.numel() / 8 if relu_bits is not None else 0))
    return out


def wgrad_scratch_bytes(M, N, R, S, Cs):
    return L().ppv_conv_wgrad_scratch_bytes(M, N, R, S, Cs)


def conv_wgrad(g, x, R, S, stride, pad, scratch=None, out=None, stream=None):
    """g [B,Ho,Wo,Cout] bf16, x [B,H,W,Cin] bf16 -> dW in torch layout [Cout,Cin,R,S] f32.
    scratch: optional uint8 buffer of >= wgrad_scratch_bytes(...) (reused across convs; no zeroing needed).
    out: optional contiguous f32 [Cout,Cin,R,S] destination (e.g. a slice of a flat gradient bucket, dist_sync.GradSync).
    stream: optional torch.cuda.Stream to launch on instead of the current one (no `with torch.cuda.stream(...)` round trip: that
    context costs the host ~10 us per use, 93 uses per step in the trunk's backward); allocate `out` / `scratch` yourself then."""
    B, Ho, Wo, Cout = g.shape
    _, H, W, Cin = x.shape
    need = wgrad_scratch_bytes(B * Ho * Wo, Cout, R, S, Cin)
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty(need, dtype=torch.uint8, device=g.device)
    if out is None:
        out = torch.empty((Cout, Cin, R, S), dtype=F32, device=g.device)
    assert out.shape == (Cout, Cin, R, S) and out.dtype == F32 and out.is_contiguous()
    _timed("conv_wgrad", 2.0 * B * Ho * Wo * Cout * R * S * Cin, lambda: check(
        L().ppv_conv_wgrad(ptr(g), ptr(x), ptr(out), ptr(scratch), ptr(zero_page(g.device)), B, H, W, Cin, Ho, Wo, Cout, R, S,
                           stride, pad, stream_ptr() if stream is None else _lib.ctypes.c_void_p(stream.cuda_stream)), "ppv_conv_wgrad"),
        nbytes=(g.numel() + x.numel()) * 2.0 + out.numel() * 4.0)
    return out


def conv_wgrad_group(gs, xs, outs=None):
    """Weight gradients of up to 24 1x1 / unit-stride convolutions of ONE shape in one launch, each reduced over all its rows (no
    split-M slabs, no reduce launch: csrc/conv_wgrad_stem.hip conv_wgrad_group_kernel).  gs[i] [B,H,W,Cout] bf16, xs[i] [B,H,W,Cin] bf16
    -> list of [Cout,Cin,1,1] f32 (outs[i] when given: contiguous f32 destinations, e.g. slices of a gradient bucket)."""
    import ctypes
    P = len(gs)
    B, H, W, Cout = gs[0].shape
    Cin = xs[0].shape[-1]
    assert 1 <= P <= 24 and all(g.shape == gs[0].shape and g.is_contiguous() for g in gs)
    assert all(x.shape == (B, H, W, Cin) and x.is_contiguous() for x in xs)
    if outs is None:
        outs = [None] * P
    outs = [o if o is not None else torch.empty((Cout, Cin, 1, 1), dtype=F32, device=gs[0].device) for o in outs]
    assert all(o.shape == (Cout, Cin, 1, 1) and o.dtype == F32 and o.is_contiguous() for o in outs)
    arr = ctypes.c_void_p * P
    ga, xa, oa = arr(*[g.data_ptr() for g in gs]), arr(*[x.data_ptr() for x in xs]), arr(*[o.data_ptr() for o in outs])
    _timed("conv_wgrad", 2.0 * P * B * H * W * Cout * Cin, lambda: check(
        L().ppv_conv_wgrad_group(ga, xa, oa, P, ptr(zero_page(gs[0].device)), B, H, W, Cin, Cout, stream_ptr()), "ppv_conv_wgrad_group"),
        nbytes=P * ((gs[0].numel() + xs[0].numel()) * 2.0 + outs[0].numel() * 4.0))
    return outs


# ----------------------------------------------------------------------------- dense layers in exact f32 (csrc/gemm_f32.hip)
import os as _os
_GEMM_WS = _os.environ.get("PPV_GEMM_WS", "1") != "0"
_gemm_ws = {}


_gemm_ws_pinned = []       # scratch blocks whose address a captured hipGraph has baked in: never handed back to the allocator


def _gemm_scratch(device, nbytes):
    """Slab scratch of linear_f32, one per (device, stream), grow-only: successive GEMMs of a stream reuse it in stream order.
    A block that was used while the stream was being captured stays alive for the life of the process (a replay writes its slabs
    to that address): when a later call outgrows it, the old block is parked in `_gemm_ws_pinned` instead of being freed."""
    key = (device.index, _lib.stream_ptr().value)
    capturing = torch.cuda.is_current_stream_capturing()
    ent = _gemm_ws.get(key)
    if ent is None or ent[0].numel() < nbytes:
        if ent is not None and ent[1]:
            _gemm_ws_pinned.append(ent[0])
        ent = [torch.empty(max(nbytes, 1 << 24), dtype=torch.uint8, device=device), False]
        _gemm_ws[key] = ent
    if capturing:
        ent[1] = True
    return ent[0]


def linear_f32(x, w, bias=None, out=None):
    """out = x @ w^T (+ bias) on v_mfma_f32_16x16x4_f32 (exact f32).  x [m, K] f32 with unit column stride (rows may be strided,
    e.g. a column block of a wider buffer), w [N, K] f32 (same), out: optional f32 [m, N] with unit column stride."""
    m, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.dtype == F32 and w.dtype == F32 and x.stride(1) == 1 and w.stride(1) == 1
    if K % 16 or x.stride(0) % 4 or w.stride(0) % 4 or x.data_ptr() % 16 or w.data_ptr() % 16:
        raise ValueError("linear_f32: K must be a multiple of 16 and rows 16-byte aligned")
    # tiled kernel with the K slices combined through slabs (no atomics, no zero-fill in front of the launch, slices summed in index
    # order): csrc/gemm_f32.hip ppv_gemm_f32_ws.  PPV_GEMM_WS=0: the atomics form for every shape
    if _GEMM_WS and N % 4 == 0 and (bias is None or bias.data_ptr() % 16 == 0):
        nbytes = _lib.ctypes.c_size_t(0)
        ks = L().ppv_gemm_f32_ws_plan(m, N, K, _lib.ctypes.byref(nbytes))
        if out is None:
            out = torch.empty((m, N), dtype=F32, device=x.device)
        else:
            assert tuple(out.shape) == (m, N) and out.stride(1) == 1 and out.dtype == F32
        if ks == 1 or (out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0):
            if ks > 1 or L().ppv_gemm_f32_ksplit(m, N, K) == 1:
                ws = _gemm_scratch(x.device, nbytes.value) if ks > 1 else None
                check(L().ppv_gemm_f32_ws(ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), m, N, K, ks, ptr(ws),
                                          stream_ptr()), "ppv_gemm_f32_ws")
                return out
    ks = L().ppv_gemm_f32_ksplit(m, N, K)
    if ks > 2 and torch.are_deterministic_algorithms_enabled():
        ks = 2          # two adders into a zeroed element commute: bit-reproducible (more would leave the order of the f32 atomics open)
    if out is None:
        out = (torch.zeros if ks > 1 else torch.empty)((m, N), dtype=F32, device=x.device)
    else:
        assert tuple(out.shape) == (m, N) and out.stride(1) == 1 and out.dtype == F32
        if ks > 1:
            out.zero_()
    check(L().ppv_gemm_f32(ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), m, N, K, ks, stream_ptr()),
          "ppv_gemm_f32")
    return out


def linear_x3(x, w, bias=None, out=None):
    """out = x @ w^T (+ bias) as three bf16 MFMA products of hi / lo parts split inside the kernel (csrc/gemm_f32.hip
    ppv_gemm_bf16x3_nt: ~1e-5 of sum |x w|, 5.3x the matrix rate of linear_f32's exact-f32 MFMA).  For large NON-recurrent products
    (the decoder's vocabulary layer over all time steps and its transposed data gradient).  x [m, K], w [N, K] f32 with unit column
    stride, K % 4 == 0, rows 16-byte aligned."""
    m, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.dtype == F32 and w.dtype == F32 and x.stride(1) == 1 and w.stride(1) == 1
    if K % 4 or x.stride(0) % 4 or w.stride(0) % 4 or x.data_ptr() % 16 or w.data_ptr() % 16:
        raise ValueError("linear_x3: K must be a multiple of 4 and rows 16-byte aligned")
    if out is None:
        out = torch.empty((m, N), dtype=F32, device=x.device)
    else:
        assert tuple(out.shape) == (m, N) and out.stride(1) == 1 and out.dtype == F32
    nbytes = _lib.ctypes.c_size_t(0)
    ks = L().ppv_gemm_bf16x3_nt_plan(m, N, K, _lib.ctypes.byref(nbytes))
    if ks > 1 and (out.stride(0) % 4 or out.data_ptr() % 16 or (bias is not None and bias.data_ptr() % 16)):
        ks = 1
    ws = _gemm_scratch(x.device, nbytes.value) if ks > 1 else None
    check(L().ppv_gemm_bf16x3_nt(ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), m, N, K, ks, ptr(ws),
                                 stream_ptr()), "ppv_gemm_bf16x3_nt")
    return out


def gemm_f32_tn(a, b, out=None, x3=False):
    """a^T b on the matrix pipe: a [K, M], b [K, N] f32 with unit column stride (rows may be strided) -> [M, N].  The batched weight
    gradient g^T h of a dense layer without transposed copies.  x3=False: exact f32 (csrc/gemm_f32.hip ppv_gemm_f32_tn); x3=True: three
    bf16 products of operands split inside the kernel (ppv_gemm_bf16x3_tn: ~1e-5 of sum |a b|, 5.3x the matrix rate)."""
    plan, run, name = ((L().ppv_gemm_bf16x3_tn_plan, L().ppv_gemm_bf16x3_tn, "ppv_gemm_bf16x3_tn") if x3 else
                       (L().ppv_gemm_f32_tn_plan, L().ppv_gemm_f32_tn, "ppv_gemm_f32_tn"))
    K, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == K and a.dtype == F32 and b.dtype == F32 and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=F32, device=a.device)
    else:
        assert tuple(out.shape) == (M, N) and out.stride(1) == 1 and out.dtype == F32
    nbytes = _lib.ctypes.c_size_t(0)
    ks = plan(M, N, K, _lib.ctypes.byref(nbytes))
    if ks > 1 and (out.stride(0) % 4 or out.data_ptr() % 16):
        ks = 1
    ws = _gemm_scratch(a.device, nbytes.value) if ks > 1 else None
    check(run(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0), M, N, K, ks, ptr(ws), stream_ptr()), name)
    return out


# ----------------------------------------------------------------------------- stem
def stem_weight_layout(w, mode):
    out = torch.empty((64, 24, 8) if mode == 0 else (16, 4, 4, 64), dtype=BF16, device=w.device)
    check(L().ppv_stem_weight_layout(ptr(w.contiguous()), ptr(out), mode, stream_ptr()), "ppv_stem_weight_layout")
    return out


def stem_conv(img, wst, stat_part=None):
    """img [B,3,H,W] f32 NCHW -> raw [B,H/2,W/2,64] bf16."""
    B, _, H, W = img.shape
    out = torch.empty((B, H // 2, W // 2, 64), dtype=BF16, device=img.device)
    check(L().ppv_stem_conv(ptr(img), ptr(wst), ptr(out), ptr(stat_part), 0 if stat_part is None else stat_part.shape[0],
                            B, H, W, stream_ptr()), "ppv_stem_conv")
    return out


def stem_dgrad(g_raw, wsd):
    """g_raw [B,Ho,Wo,64] bf16 -> d/d(img) [B,3,2Ho,2Wo] f32 NCHW."""
    B, Ho, Wo, _ = g_raw.shape
    if Wo % 128 == 0:                      # row-staged single launch (the 256^2 workload)
        out = torch.empty((B, 3, 2 * Ho, 2 * Wo), dtype=F32, device=g_raw.device)
        check(L().ppv_stem_dgrad(ptr(g_raw), ptr(wsd), ptr(out), ptr(zero_page(g_raw.device)), B, Ho, Wo, stream_ptr()), "ppv_stem_dgrad")
        return out
    tmp = torch.empty((B * Ho * Wo, 16), dtype=F32, device=g_raw.device)
    check(L().ppv_conv_gemm(ptr(g_raw), ptr(wsd), ptr(tmp), None, None, None, ptr(zero_page(g_raw.device)), B, Ho, Wo, 64,
                            Ho, Wo, 16, 4, 4, 1, -1, 1, 1, 0, stream_ptr()), "ppv_conv_gemm(stem dgrad)")
    out = torch.empty((B, 3, 2 * Ho, 2 * Wo), dtype=F32, device=g_raw.device)
    check(L().ppv_stem_dgrad_scatter(ptr(tmp), ptr(out), B, Ho, Wo, stream_ptr()), "ppv_stem_dgrad_scatter")
    return out


# ----------------------------------------------------------------------------- batch norm
def bn_finalize(stat_part, count, gamma, beta, run_mean, run_var, momentum=0.1, eps=1e-5):
    """-> coef [4,C] f32 = scale, shift, mean, invstd; running stats updated in place (may be None)."""
    T, _, C = stat_part.shape
    coef = torch.empty((4, C), dtype=F32, device=stat_part.device)
    check(L().ppv_bn_finalize(ptr(stat_part), T, float(count), ptr(gamma), ptr(beta), ptr(run_mean), ptr(run_var),
                              momentum, eps, ptr(coef), C, stream_ptr()), "ppv_bn_finalize")
    return coef


def bn_act_train(x, stat_part, count, bn, momentum, res=None, res_stats=None, relu=True, want_bits=False):
    """Train-mode BatchNorm2d `bn` on the raw conv output x [..., C] bf16 from its partial statistics stat_part [T,2,C] (+ ReLU, +
    residual) in ONE launch (csrc/trunk_ops.hip bn_act_train_kernel = bn_finalize + bn_act).  res: identity residual, or with
    res_stats = (stat_part2, bn2, momentum2) the raw output of a projection shortcut normalised by its own BatchNorm.
    -> (y, bits or None, coef [4,C], coef2 or None); running statistics of bn (and bn2) updated in place."""
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    bits = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device) if want_bits else None
    coef = torch.empty((4, C), dtype=F32, device=x.device)
    coef2 = None
    a2 = [None, 0, None, None, None, None, 0.0, 0.0, None]
    mode = 0 if res is None else 1
    if res_stats is not None:
        sp2, bn2, mom2 = res_stats
        coef2 = torch.empty((4, C), dtype=F32, device=x.device)
        a2 = [ptr(sp2), sp2.shape[0], ptr(bn2.weight.detach()), ptr(bn2.bias.detach()), ptr(bn2.running_mean), ptr(bn2.running_var),
              mom2, bn2.eps, ptr(coef2)]
        mode = 2
    check(L().ppv_bn_act_train(ptr(x), ptr(stat_part), stat_part.shape[0], float(count), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                               ptr(bn.running_mean), ptr(bn.running_var), momentum, bn.eps, ptr(coef), ptr(res), *a2, ptr(y), ptr(bits),
                               rows, C, mode, int(relu), stream_ptr()), "ppv_bn_act_train")
    return y, bits, coef, coef2


def bn_act_fold(x, sums, count, bn, momentum, res=None, relu=True, want_bits=False):
    """Train-mode BatchNorm2d `bn` (+ identity residual) (+ ReLU) on the raw conv output x [..., C] bf16 whose statistics sit in ONE row
    sums [T,2,C] (conv_fwd(..., stat_part=sums) with T partial rows, T = 1 .. 8): no bn_finalize launch, every thread derives its channels'
    coefficients (csrc/trunk_ops.hip bn_act_fold_kernel).  -> (y, bits or None, coef [4,C]); running statistics updated in place."""
    C = x.shape[-1]
    y = torch.empty_like(x)
    bits = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device) if want_bits else None
    coef = torch.empty((4, C), dtype=F32, device=x.device)
    check(L().ppv_bn_act_fold_rows(ptr(x), ptr(sums), sums.shape[0], float(count), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                                   ptr(bn.running_mean), ptr(bn.running_var), momentum, bn.eps, ptr(coef), ptr(res), ptr(y), ptr(bits),
                                   x.numel(), C, 0 if res is None else 1, int(relu), stream_ptr()), "ppv_bn_act_fold_rows")
    return y, bits, coef


def conv3x3_bnin_supported(B, H, W, C, N):
    return bool(L().ppv_conv3x3_bnin_supported(B, H, W, C, N))


def conv3x3_bnin(x_raw, sums, count, bn, momentum, wt, stat_part=None, want_act=True):
    """conv3x3(relu(bn(x_raw))) with the train-mode BatchNorm `bn` + ReLU applied INSIDE the convolution (csrc/conv_halo.hip BNIN, round 6):
    x_raw [B,H,W,C] bf16 raw output of the previous convolution, sums [T,2,C] its partial sums, wt [N,3,3,C] bf16.
    -> (out [B,H,W,N] bf16 raw, y_act [B,H,W,C] bf16 or None, coef [4,C]); running statistics updated in place; stat_part [rows,2,N] PRE-ZEROED."""
    B, H, W, C = x_raw.shape
    N = wt.shape[0]
    out = torch.empty((B, H, W, N), dtype=BF16, device=x_raw.device)
    y = torch.empty_like(x_raw) if want_act else None
    coef = torch.empty((4, C), dtype=F32, device=x_raw.device)
    _timed("conv3x3_bnin", 2.0 * B * H * W * N * 9 * C, lambda: check(
        L().ppv_conv3x3_bnin(ptr(x_raw), ptr(sums), sums.shape[0], float(count), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                             ptr(bn.running_mean), ptr(bn.running_var), momentum, bn.eps, ptr(coef), ptr(y), ptr(wt), ptr(out), ptr(stat_part),
                             0 if stat_part is None else stat_part.shape[0], ptr(zero_page(x_raw.device)), B, H, W, C, N, stream_ptr()),
        "ppv_conv3x3_bnin"), nbytes=x_raw.numel() * 2.0 * (2 if want_act else 1) + wt.numel() * 2.0 + out.numel() * 2.0)
    return out, y, coef


def bn_act(x, coef, res=None, coef_res=None, relu=True, res_broadcast=False, want_bits=False):
    """y = act(x*scale + shift + res); res_broadcast: res holds one image's worth of elements shared by the batch.
    want_bits: also return the (y > 0) bit mask (uint8, numel / 8 bytes) that conv_dgrad(relu_bits=...) consumes."""
    y = torch.empty_like(x)
    bits = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device) if want_bits else None
    mode = 0 if res is None else (1 if coef_res is None else 2)
    check(L().ppv_bn_act(ptr(x), ptr(coef), ptr(res), ptr(coef_res), ptr(y), ptr(bits), x.numel(), x.shape[-1], mode, int(relu),
                         res.numel() if (res is not None and res_broadcast) else 0, stream_ptr()), "ppv_bn_act")
    return (y, bits) if want_bits else y


# BatchNorm backward in EVAL mode (running statistics are constants): g_x = scale * g (.) mask, d gamma = sum g x_hat, d beta = sum g.
# The kernels compute train-mode g_x = k0 g + k1 x + k2 with k1, k2 proportional to 1 / count (csrc/trunk_ops.hip): an infinite
# sample count makes both exactly zero while d gamma / d beta (which do not contain the count) stay right.  Set by the trunk's backward
# around an eval-mode pass (encoder._TrunkFn.backward).
# Thread-local (autograd runs backward on per-device engine threads: a process-global flag would leak into a concurrent train-mode
# backward of another encoder / device); use `with eval_bn():` around the pass.
import contextlib as _contextlib
import threading as _threading

_BN_TLS = _threading.local()


@_contextlib.contextmanager
def eval_bn():
    prev = getattr(_BN_TLS, "eval", False)
    _BN_TLS.eval = True
    try:
        yield
    finally:
        _BN_TLS.eval = prev


def _bn_count(rows):
    return float("inf") if getattr(_BN_TLS, "eval", False) else float(rows)


def bn_bwd(gy, y, x, coef, relu, want_gpre=False, want_affine=True, part=None, part_ready=False, sums2=None, out_affine=None):
    """-> (g_x bf16, g_pre bf16|None, dgamma f32|None, dbeta f32|None).  part: optional PRE-ZEROED f32 [64*C] scratch;
    part_ready: part already holds the sums (conv_dgrad(..., red=(x, part)) produced gy).  sums2 = (x2, part2): also take the
    backward sums of a second BatchNorm that the same gy feeds (raw output x2) into the PRE-ZEROED part2."""
    C = x.shape[-1]
    rows = x.numel() // C
    dev = x.device
    gx = torch.empty_like(x)
    gpre = torch.empty_like(x) if want_gpre else None
    dg = torch.empty(C, dtype=F32, device=dev) if want_affine else None
    db = torch.empty(C, dtype=F32, device=dev) if want_affine else None
    if want_affine and out_affine is not None:          # (d gamma, d beta) destinations, e.g. slices of a flat gradient bucket
        dg, db = out_affine
    prezeroed = part is not None
    if part is None:
        part = torch.empty(64 * C, dtype=F32, device=dev)
    kc = torch.empty(3 * C, dtype=F32, device=dev)
    if sums2 is not None:               # (x2, part2): the projection shortcut's BN sums ride along (relu 0, no g_pre copy)
        assert not relu and not want_gpre
        check(L().ppv_bn_bwd_sums2(ptr(gy), ptr(x), ptr(coef), _bn_count(rows), ptr(gx), ptr(dg), ptr(db), ptr(part), ptr(kc), rows, C,
                                   2 if part_ready else int(prezeroed), ptr(sums2[0]), ptr(sums2[1]), stream_ptr()), "ppv_bn_bwd_sums2")
        return gx, gpre, dg, db
    check(L().ppv_bn_bwd(ptr(gy), ptr(y), ptr(x), ptr(coef), _bn_count(rows), ptr(gx), ptr(gpre), ptr(dg), ptr(db), ptr(part),
                         ptr(kc), rows, C, int(relu), 2 if part_ready else int(prezeroed), stream_ptr()), "ppv_bn_bwd")
    return gx, gpre, dg, db


# ----------------------------------------------------------------------------- pools
def bn_relu_maxpool(x, coef):
    B, H, W, C = x.shape
    y = torch.empty((B, H // 2, W // 2, C), dtype=BF16, device=x.device)
    arg = torch.empty((B, H // 2, W // 2, C), dtype=torch.uint8, device=x.device)
    check(L().ppv_bn_relu_maxpool(ptr(x), ptr(coef), ptr(y), ptr(arg), B, H, W, C, stream_ptr()), "ppv_bn_relu_maxpool")
    return y, arg


def maxpool_relu_bwd(gy, y, arg, in_hw):
    B, Ho, Wo, C = gy.shape
    H, W = in_hw
    g = torch.empty((B, H, W, C), dtype=BF16, device=gy.device)
    check(L().ppv_maxpool_relu_bwd(ptr(gy), ptr(y), ptr(arg), ptr(g), B, H, W, C, stream_ptr()), "ppv_maxpool_relu_bwd")
    return g


def maxpool_bn_bwd(gy, y, arg, x_raw, coef, want_affine=False):
    """Backward of BatchNorm(train) + ReLU + MaxPool 3x3/2 from the pooled gradient gy [B,Ho,Wo,C] to the gradient of the raw conv
    output x_raw [B,2Ho,2Wo,C] (two passes over the pooled tensors + x_raw: no pre-pool gradient tensor).
    -> (g_x bf16, dgamma f32|None, dbeta f32|None)."""
    B, H, W, C = x_raw.shape
    gx = torch.empty_like(x_raw)
    dg = torch.empty(C, dtype=F32, device=x_raw.device) if want_affine else None
    db = torch.empty(C, dtype=F32, device=x_raw.device) if want_affine else None
    part = torch.empty(16 * C, dtype=F32, device=x_raw.device)
    check(L().ppv_maxpool_bn_bwd(ptr(gy), ptr(y), ptr(arg), ptr(x_raw), ptr(coef), _bn_count(B * H * W), ptr(gx), ptr(dg), ptr(db), ptr(part),
                                 B, H, W, C, stream_ptr()), "ppv_maxpool_bn_bwd")
    return gx, dg, db


def adaptive_pool_fwd(x, E, out_dtype=F32, out=None):
    B, H, W, C = x.shape
    y = torch.empty((B, E, E, C), dtype=out_dtype, device=x.device) if out is None else out
    assert y.shape == (B, E, E, C) and y.dtype == out_dtype and y.is_contiguous()
    check(L().ppv_adaptive_pool_fwd(ptr(x), ptr(y), B, H, W, C, E, int(out_dtype == F32), stream_ptr()), "ppv_adaptive_pool_fwd")
    return y


def adaptive_pool_bwd(gy, in_hw, relu_of=None):
    """relu_of: the pooled tensor (a ReLU output) -> its mask is applied to the returned gradient."""
    B, E, _, C = gy.shape
    H, W = in_hw
    gx = torch.empty((B, H, W, C), dtype=BF16, device=gy.device)
    check(L().ppv_adaptive_pool_bwd(ptr(gy.contiguous()), ptr(gx), ptr(relu_of), B, H, W, C, E, int(gy.dtype == F32), stream_ptr()),
          "ppv_adaptive_pool_bwd")
    return gx
