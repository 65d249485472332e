"""Differentiable building blocks on the HIP kernels, NHWC float32 -- for the parts of the reference that TRAIN through fp32
convolutions outside the bf16 trunk: FAN's ``get_heatmap_train`` (core/wing.py:262-272: the landmark loss back-propagates through
the frozen regressor into the generated image) and the StarGAN-v2 blocks (core/model.py:12-124).

``conv2d_f32``  convolution as three bf16 MFMA products accumulated in f32 ([x_hi | x_lo | x_hi] x [W_hi | W_hi | W_lo], product
                error ~2^-16), forward AND data gradient (the same split on the incoming gradient against the flipped weights);
                weight gradient on the bf16 MFMA weight-gradient kernel: directly where its tile rules hold (channel counts
                multiples of 128), otherwise on zero-padded bf16-split operands / K-padded patch rows (``_wgrad_padded``).
``instance_norm_act``  InstanceNorm2d / AdaIN + LeakyReLU, forward and backward (csrc/instnorm.hip)."""
import weakref

import torch
import torch.nn.functional as F

from . import _lib, convops as co
from ._lib import check, ptr, stream_ptr

_wcache = {}            # id(weight tensor) -> (version key, forward layout, data-gradient layout, weakref to that tensor)


def _pad_to(t, dim, mult):
    n = t.shape[dim]
    p = (n + mult - 1) // mult * mult - n
    if p == 0:
        return t
    pad = [0, 0] * (t.dim() - 1 - dim) + [0, p]
    return F.pad(t, pad)


def _layouts(w, exact=False):
    """(forward GEMM rows over the split input, data-gradient GEMM rows over the split gradient) of conv weight [Cout,Cin,R,S].
    Cached per weight OBJECT: an entry lives exactly as long as the tensor it was built from (its weak reference drops the entry when
    the tensor dies, so a recycled id() can never return another tensor's layout and layouts of dead models are released) and is
    rebuilt when the tensor's version counter or storage changes.  Pass the long-lived Parameter (conv2d_f32(..., weight_grad=False)
    for frozen weights), not a `.detach()` temporary: a temporary's entry dies with it, i.e. the layouts are rebuilt on every call.
    In-place writes through ``.data`` bump no version counter: call ``drop_layouts(w)`` after such a write."""
    key = id(w)
    ent = _wcache.get(key)
    ver = (w._version, w.data_ptr(), w.device)
    if ent is None or ent[0] != ver or ent[3]() is not w or (exact and ent[4] is None):
        wf = w.detach().float()
        hi = wf.bfloat16().float()
        lo = (wf - hi).bfloat16().float()
        fwd = _pad_to(_pad_to(torch.cat([hi, hi, lo], dim=1), 1, 64), 0, 64)                # [Coutp, pad64(3 Cin), R, S]
        bwd = _pad_to(_pad_to(torch.cat([hi, hi, lo], dim=0), 0, 64), 1, 64)                # [pad64(3 Cout), Cinp, R, S]
        second = None
        if exact:
            # all six products of a three-way split that matter (x = h + m + l, W = H + M + L; `lo` above is M) in ONE convolution:
            # [x_h | x_m | x_h | x_l | x_h | x_m] (ppv_split6_rows) against the K-concatenated filter [W_H | W_H | W_M | W_H | W_L | W_M]
            l3 = (wf - hi - lo).bfloat16().float()
            fwd6 = _pad_to(_pad_to(torch.cat([hi, hi, lo, hi, l3, lo], dim=1), 1, 64), 0, 64)
            bwd6 = _pad_to(_pad_to(torch.cat([hi, hi, lo, hi, l3, lo], dim=0), 0, 64), 1, 64)
            second = (co.weight_layout(fwd6.contiguous(), 0), co.weight_layout(bwd6.contiguous(), 1))
        ent = (ver, co.weight_layout(fwd.contiguous(), 0), co.weight_layout(bwd.contiguous(), 1),
               weakref.ref(w, lambda _r, k=key: _wcache.pop(k, None)), second)
        _wcache[key] = ent
    if exact:
        return ent[1], ent[2], ent[4][0], ent[4][1]
    return ent[1], ent[2]


def drop_layouts(w=None):
    """Forget the cached layouts of ``w`` (all weights when None)."""
    if w is None:
        _wcache.clear()
    else:
        _wcache.pop(id(w), None)


def _split3(t):
    B, H, W, C = t.shape
    cp = (3 * C + 63) // 64 * 64
    y = torch.empty((B, H, W, cp), dtype=torch.bfloat16, device=t.device)
    check(_lib.lib().ppv_bn_act_split3(ptr(t), None, ptr(y), B * H * W, C, cp, 0, C, stream_ptr()), "ppv_bn_act_split3")
    return y


def _split6(t):
    """[h | m | h | l | h | m] of the three-way bf16 split t = h + m + l (h = bf16(t), m = bf16(t - h), l = bf16(t - h - m)), zero padded to a
    multiple of 64 channels: the operand of conv2d_f32(exact=True) (csrc/bn_f32.hip split6_kernel; C % 4 == 0)."""
    B, H, W, C = t.shape
    if C % 4:
        t = _pad_to(t, 3, 4).contiguous()
        C = t.shape[-1]
    cp = (6 * C + 63) // 64 * 64
    y = torch.empty((B, H, W, cp), dtype=torch.bfloat16, device=t.device)
    check(_lib.lib().ppv_split6_rows(ptr(t), ptr(y), B * H * W, C, cp, stream_ptr()), "ppv_split6_rows")
    return y


def _wgrad_padded(gy, x, wshape, stride, pad, exact=False):
    """Weight gradient [Cout,Cin,R,S] f32 of an NHWC f32 convolution with ANY channel counts on the MFMA weight-gradient kernel
    (ppv_conv_wgrad wants multiples of 128): both operands as bf16-split halves stacked along the batch, x -> [hi; lo; hi], gy ->
    [hi; hi; lo] (the sum over the batch then holds x_hi g_hi + x_lo g_hi + x_hi g_lo: the f32 product to ~2^-16), zero-padded to 128
    channels.  Very few input channels (the RGB convolutions) go through their R x S patch rows instead (csrc/im2col.hip): K = R S Cin
    padded to 128 and a 1x1 weight gradient -- padding 3 channels to 128 per tap would move 40x the bytes."""
    Cout, Cin, R, S = wshape
    B, H, W, _ = x.shape
    _, Ho, Wo, _ = gy.shape
    L = _lib.lib()
    dev = x.device
    Np = (Cout + 127) // 128 * 128
    # exact: all six products of the three-way split, x -> [h; m; h; l; h; m] against gy -> [h; h; m; h; l; m] (~2^-24 per product)
    gparts, xparts = ((0, 0, 1, 0, 2, 1), (0, 1, 0, 2, 0, 1)) if exact else ((0, 0, 1), (0, 1, 0))
    T = len(gparts)
    gs = torch.empty((T * B, Ho, Wo, Np), dtype=torch.bfloat16, device=dev)
    rows_g = B * Ho * Wo
    for j, part in enumerate(gparts):
        check(L.ppv_pad_split(ptr(gy), gs[j * B].data_ptr(), rows_g, Cout, Np, part, stream_ptr()), "ppv_pad_split")
    if Cin < 32 and R * S * Cin <= 512:
        Kp = (R * S * Cin + 127) // 128 * 128
        xs = torch.empty((T * B, Ho, Wo, Kp), dtype=torch.bfloat16, device=dev)
        for j, part in enumerate(xparts):
            check(L.ppv_im2col_split(ptr(x), xs[j * B].data_ptr(), B, H, W, Cin, Ho, Wo, R, S, stride, pad, Kp, part, stream_ptr()),
                  "ppv_im2col_split")
        gw = co.conv_wgrad(gs, xs, 1, 1, 1, 0)                                   # [Np, Kp, 1, 1]
        return gw.view(Np, Kp)[:Cout, :R * S * Cin].reshape(Cout, R, S, Cin).permute(0, 3, 1, 2).contiguous()
    Cp = (Cin + 127) // 128 * 128
    xs = torch.empty((T * B, H, W, Cp), dtype=torch.bfloat16, device=dev)
    for j, part in enumerate(xparts):
        check(L.ppv_pad_split(ptr(x), xs[j * B].data_ptr(), B * H * W, Cin, Cp, part, stream_ptr()), "ppv_pad_split")
    gw = co.conv_wgrad(gs, xs, R, S, stride, pad)                                # [Np, Cp, R, S]
    return gw[:Cout, :Cin].contiguous()


class _ConvF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, wf, wd, accurate_wgrad=False, second=None):
        _lib.require_cuda(x, weight)
        ctx.accurate_wgrad = accurate_wgrad
        x = x.contiguous().float()
        Cout, Cin, R, S = weight.shape
        assert R == S and x.shape[-1] == Cin
        ctx.wd = wd
        ctx.wd2 = second[1] if second is not None else None
        if second is not None:                       # exact: six products of the three-way split, one K-concatenated convolution
            y = co.conv_fwd(_split6(x), second[0], stride, pad, out_f32=True)
        else:
            y = co.conv_fwd(_split3(x), wf, stride, pad, out_f32=True)
        if y.shape[-1] != Cout:
            y = y[..., :Cout].contiguous()
        if bias is not None:
            y = y + bias.detach().float()
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, pad, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, pad, has_bias = ctx.geom
        Cout, Cin, R, S = weight.shape
        gy = gy.contiguous().float()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if ctx.wd2 is not None:
                gx = co.conv_dgrad(_split6(gy), ctx.wd2, stride, pad, (x.shape[1], x.shape[2]), out_f32=True)
            else:
                gx = co.conv_dgrad(_split3(gy), ctx.wd, stride, pad, (x.shape[1], x.shape[2]), out_f32=True)
            if gx.shape[-1] != Cin:
                gx = gx[..., :Cin].contiguous()
        if ctx.needs_input_grad[1]:
            if Cout % 128 == 0 and Cin % 128 == 0 and not ctx.accurate_wgrad:
                gw = co.conv_wgrad(gy.bfloat16(), x.bfloat16(), R, S, stride, pad)
            else:                                       # outside the MFMA weight-gradient kernel's tiles (3- and 64-channel layers)
                gw = _wgrad_padded(gy, x, weight.shape, stride, pad, exact=ctx.wd2 is not None)
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(dim=(0, 1, 2))
        return gx, gw, gb, None, None, None, None, None, None


def conv2d_f32(x, weight, bias=None, stride=1, pad=0, weight_grad=True, accurate_wgrad=False, exact=False):
    """x [B,H,W,Cin] f32 NHWC, weight [Cout,Cin,k,k] (torch layout, e.g. an nn.Conv2d's parameter) -> [B,Ho,Wo,Cout] f32.
    weight_grad=False: the weight (and bias) are treated as constants whatever their requires_grad says (a frozen network, FAN in
    core/wing.py:262-272) -- the bf16 layouts stay cached on the Parameter object itself.  accurate_wgrad: the weight gradient from
    the bf16-split operands (~2^-16) for every shape (default: one bf16 product where the channel counts fit the kernel's tiles)."""
    if exact:
        # reference precision: a THREE-way bf16 split of both operands and all six products that matter (h H, m H, h M, l H, h L, m M;
        # the dropped ones are <= 2^-32 of the product), i.e. ~2^-24 per product -- f32 -- instead of 2^-16, as ONE convolution over the
        # K-concatenated operands [x_h | x_m | x_h | x_l | x_h | x_m] x [W_H | W_H | W_M | W_H | W_L | W_M]; forward, data gradient and
        # weight gradient (the fp32 product mode of ppv_amd.encoder.Encoder: 2x the work of the three-product form)
        wf, wd, wf2, wd2 = _layouts(weight, exact=True)
        second = (wf2, wd2)
        accurate_wgrad = True
    else:
        wf, wd = _layouts(weight)
        second = None
    if not weight_grad:
        weight = weight.detach()
        bias = None if bias is None else bias.detach()
    return _ConvF32.apply(x, weight, bias, stride, pad, wf, wd, accurate_wgrad, second)


class _InstNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, slope, eps):
        _lib.require_cuda(x)
        x = x.contiguous().float()
        B, H, W, C = x.shape
        per_sample = scale.dim() == 2
        scale, shift = scale.contiguous().float(), shift.contiguous().float()
        y = torch.empty_like(x)
        stats = torch.empty((B, C, 2), dtype=torch.float32, device=x.device)
        check(_lib.lib().ppv_instnorm_fwd(ptr(x), ptr(scale), ptr(shift), ptr(y), ptr(stats), B, H * W, C, int(per_sample), slope, eps,
                                          stream_ptr()), "ppv_instnorm_fwd")
        ctx.save_for_backward(x, stats, scale, shift)
        ctx.cfg = (per_sample, slope)
        return y

    @staticmethod
    def backward(ctx, g):
        x, stats, scale, shift = ctx.saved_tensors
        per_sample, slope = ctx.cfg
        B, H, W, C = x.shape
        g = g.contiguous().float()
        dx = torch.empty_like(x)
        sums = torch.empty((B, C, 2), dtype=torch.float32, device=x.device)
        check(_lib.lib().ppv_instnorm_bwd(ptr(x), ptr(g), ptr(stats), ptr(scale), ptr(shift), ptr(dx), ptr(sums), B, H * W, C,
                                          int(per_sample), slope, stream_ptr()), "ppv_instnorm_bwd")
        dshift, dscale = sums[..., 0], sums[..., 1]
        if not per_sample:
            dshift, dscale = dshift.sum(0), dscale.sum(0)
        return dx, dscale, dshift, None, None


def instance_norm_act(x, scale, shift, slope=1.0, eps=1e-5):
    """lrelu_slope( InstanceNorm(x) * scale + shift ) on x [B,H,W,C] f32; scale / shift [C] (affine InstanceNorm2d) or [B,C] (AdaIN)."""
    return _InstNormAct.apply(x, scale, shift, float(slope), float(eps))


# ------------------------------------------------------------------------------------------------------------------------------------
# fp32 trunk, element-wise side (csrc/bn_f32.hip, round 6): what Encoder(precision="fp32") puts between the f32 convolutions

_BN_WS = {}


def _bn_ws(dev, C):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, C)
    if key not in _BN_WS:
        _BN_WS[key] = torch.empty(int(_lib.lib().ppv_bn_f32_workspace_bytes(C)), dtype=torch.uint8, device=dev)
    return _BN_WS[key]


class _BatchNormF32(torch.autograd.Function):
    """train-mode (or eval-mode) BatchNorm2d (+ residual) (+ ReLU) on NHWC f32; running statistics are updated by the forward kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, bn, relu):
        if not x.is_cuda:
            raise RuntimeError("ppv_amd batch_norm_f32 runs on an MI355X (cuda tensors); no CPU path")
        x = x.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        train = bool(bn.training or bn.running_mean is None)
        y = torch.empty_like(x)
        coef = torch.empty((4, C), dtype=torch.float32, device=x.device)
        r = None if res is None else res.contiguous()
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        track = train and bn.track_running_stats and bn.running_mean is not None
        if track and bn.momentum is None:                        # cumulative moving average (nn.BatchNorm2d(momentum=None))
            mom = 1.0 / float(int(bn.num_batches_tracked) + 1)
        check(_lib.lib().ppv_bn_f32_fwd(ptr(x), ptr(weight.detach()), ptr(bias.detach()), ptr(bn.running_mean) if (track or not train) else None,
                                        ptr(bn.running_var) if (track or not train) else None, mom, float(bn.eps), ptr(r), ptr(y), ptr(coef),
                                        ptr(_bn_ws(x.device, C)), rows, C, int(relu), int(train), stream_ptr()), "ppv_bn_f32_fwd")
        if track and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        ctx.save_for_backward(x, y if relu else None, coef)
        ctx.relu, ctx.train, ctx.has_res = relu, train, res is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, coef = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        g = g.contiguous()
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if ctx.has_res else None
        want_affine = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dg = torch.empty(C, dtype=torch.float32, device=x.device) if want_affine else None
        db = torch.empty(C, dtype=torch.float32, device=x.device) if want_affine else None
        sums = torch.empty((2, C), dtype=torch.float32, device=x.device)
        check(_lib.lib().ppv_bn_f32_bwd(ptr(g), ptr(y), ptr(x), ptr(coef), ptr(gx), ptr(gres), ptr(dg), ptr(db), ptr(sums), ptr(_bn_ws(x.device, C)),
                                        rows, C, int(ctx.relu), int(ctx.train), stream_ptr()), "ppv_bn_f32_bwd")
        return (gx if ctx.needs_input_grad[0] else None, dg if ctx.needs_input_grad[1] else None, db if ctx.needs_input_grad[2] else None,
                gres if (ctx.has_res and ctx.needs_input_grad[3]) else None, None, None)


def batch_norm_f32(x, bn, res=None, relu=True):
    """act(bn(x) + res) for an nn.BatchNorm2d module `bn` on NHWC f32 `x` [B,H,W,C] (C % 64 == 0): batch statistics + running-statistics update
    in train mode, running statistics in eval mode; gradients to x, bn.weight, bn.bias and res."""
    return _BatchNormF32.apply(x, bn.weight, bn.bias, res, bn, relu)


class _MaxPoolF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, H, W, C = x.shape
        y = torch.empty((B, (H + 1) // 2, (W + 1) // 2, C), dtype=torch.float32, device=x.device)
        arg = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
        check(_lib.lib().ppv_maxpool_f32_fwd(ptr(x), ptr(y), ptr(arg), B, H, W, C, stream_ptr()), "ppv_maxpool_f32_fwd")
        ctx.save_for_backward(arg)
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, g):
        arg, = ctx.saved_tensors
        B, H, W, C = ctx.shape
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        check(_lib.lib().ppv_maxpool_f32_bwd(ptr(g.contiguous()), ptr(arg), ptr(gx), B, H, W, C, stream_ptr()), "ppv_maxpool_f32_bwd")
        return gx


def max_pool3x3s2_f32(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (resnet.3) on NHWC f32."""
    return _MaxPoolF32.apply(x)


class _AdaptivePoolF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, E):
        x = x.contiguous()
        B, H, W, C = x.shape
        y = torch.empty((B, E, E, C), dtype=torch.float32, device=x.device)
        check(_lib.lib().ppv_adaptive_pool_f32_fwd(ptr(x), ptr(y), B, H, W, C, E, stream_ptr()), "ppv_adaptive_pool_f32_fwd")
        ctx.shape, ctx.E = (B, H, W, C), E
        return y

    @staticmethod
    def backward(ctx, g):
        B, H, W, C = ctx.shape
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        check(_lib.lib().ppv_adaptive_pool_f32_bwd(ptr(g.contiguous()), ptr(gx), B, H, W, C, ctx.E, stream_ptr()), "ppv_adaptive_pool_f32_bwd")
        return gx, None


def adaptive_avg_pool_f32(x, E):
    """nn.AdaptiveAvgPool2d((E, E)) (models.py:27) on NHWC f32 [B,H,W,C] -> [B,E,E,C] (already the layout models.py:40 permutes to)."""
    return _AdaptivePoolF32.apply(x, E)


class _BilinearResize(torch.autograd.Function):
    """F.interpolate(x, mode='bilinear', align_corners=...) on NCHW f32 with torch's source-index arithmetic, forward and adjoint on
    csrc/fan.hip (ppv_bilinear_resize_fwd / _bwd: the adjoint is a deterministic gather).  Reference use: FAN.get_heatmap_train,
    Face-DeId/core/wing.py:264 (input to 256 x 256) and :270 (heat-maps x4, align_corners=True)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo, align_corners):
        from ._lib import lib, check, ptr, stream_ptr
        xc = x.detach().float().contiguous()
        B, C, Hi, Wi = xc.shape
        y = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
        check(lib().ppv_bilinear_resize_fwd(ptr(xc), ptr(y), B * C, Hi, Wi, Ho, Wo, int(bool(align_corners)), stream_ptr()), "ppv_bilinear_resize_fwd")
        ctx.geom = (B, C, Hi, Wi, Ho, Wo, int(bool(align_corners)), x.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        from ._lib import lib, check, ptr, stream_ptr
        B, C, Hi, Wi, Ho, Wo, ac, dt = ctx.geom
        g = gy.detach().float().contiguous()
        gx = torch.empty((B, C, Hi, Wi), dtype=torch.float32, device=gy.device)
        check(lib().ppv_bilinear_resize_bwd(ptr(g), ptr(gx), B * C, Hi, Wi, Ho, Wo, ac, stream_ptr()), "ppv_bilinear_resize_bwd")
        return gx.to(dt), None, None, None


def bilinear_resize(x, size=None, scale_factor=None, align_corners=False):
    """torch.nn.functional.interpolate(x, size= / scale_factor=, mode='bilinear', align_corners=) for 4-D NCHW tensors on the HIP kernels
    (output size floor(in * scale_factor) as torch; the sampling scale is in / out resp. (in - 1) / (out - 1), i.e. torch's behaviour
    for `size=` and for align_corners=True -- the two forms the reference uses)."""
    if x.dim() != 4:
        raise ValueError("bilinear_resize: [B, C, H, W] tensors")
    if (size is None) == (scale_factor is None):
        raise ValueError("bilinear_resize: exactly one of size / scale_factor")
    if size is not None:
        Ho, Wo = (size, size) if isinstance(size, int) else tuple(size)
    else:
        if not align_corners and float(scale_factor) != int(scale_factor):
            raise ValueError("bilinear_resize: fractional scale_factor with align_corners=False samples with 1 / scale_factor in torch; pass size=")
        Ho, Wo = int(x.shape[2] * scale_factor), int(x.shape[3] * scale_factor)
    return _BilinearResize.apply(x, int(Ho), int(Wo), bool(align_corners))
