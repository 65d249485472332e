"""Drop-in for reference ``Face-DeId/core/wing.py:178 FAN`` (forward / ``get_heatmap``, eval mode) on MI355X.

Same module tree and ``state_dict`` names (``conv1.conv.weight``, ``conv2.bn1.*``, ``m0.b1_4.conv1.weight``,
``top_m_0.*``, ``conv_last0.*``, ``bn_end0.*``, ``l0.*``) so the reference's pretrained ``wing.ckpt`` loads unchanged
(``load_pretrained_weights``).  The ``torch.nn`` sub-modules are PARAMETER HOLDERS; ``forward`` runs the whole network
through libppv_hip.so on NHWC bfloat16 activations (fp32 accumulation): the 6-channel CoordConv stem, every 3x3 / 1x1
convolution on MFMA, eval-mode BatchNorm + ReLU as a pre-activation pass, pooling / up-sampling / concatenation and the
heat-map head as fused element-wise kernels.  The reference only ever calls it under ``no_grad`` at 256 x 256
(``get_heatmap`` is decorated, wing.py:240); no autograd graph is attached.
"""
from functools import partial

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from . import convops as co
from ._lib import check, ptr, stream_ptr


def _coord_map(w_tail, coords):
    """sum_c w_tail[o, c] * coords[c, h, w] -> [h, w, o]: the three CoordConv coordinate channels' contribution (a 3-term sum per output,
    built once per weight load; element-wise, no library GEMM)."""
    return (coords.permute(1, 2, 0).unsqueeze(-1) * w_tail.t().reshape(1, 1, w_tail.shape[1], w_tail.shape[0])).sum(2)


def _coord_channels(h, w):
    xx = (torch.arange(h).unsqueeze(1).expand(h, w).float() / (h - 1)) * 2 - 1          # wing.py:86-90
    yy = (torch.arange(w).unsqueeze(0).expand(h, w).float() / (w - 1)) * 2 - 1
    rr = torch.sqrt(xx ** 2 + yy ** 2)
    return torch.stack([xx, yy, rr / rr.max()], 0)                                      # [3,h,w]


class AddCoordsTh(nn.Module):
    def __init__(self, height=64, width=64, with_r=False, with_boundary=False):
        super().__init__()
        self.with_r, self.with_boundary = with_r, with_boundary


class CoordConvTh(nn.Module):
    def __init__(self, height, width, with_r, with_boundary, in_channels, first_one=False, *args, **kwargs):
        super().__init__()
        self.addcoords = AddCoordsTh(height, width, with_r, with_boundary)
        in_channels += 2 + (1 if with_r else 0) + (2 if (with_boundary and not first_one) else 0)
        self.conv = nn.Conv2d(in_channels=in_channels, *args, **kwargs)
        self.hw = (height, width)


class ConvBlock(nn.Module):
    def __init__(self, in_planes, out_planes):
        super().__init__()
        conv3x3 = partial(nn.Conv2d, kernel_size=3, stride=1, padding=1, bias=False, dilation=1)
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.conv1 = conv3x3(in_planes, out_planes // 2)
        self.bn2 = nn.BatchNorm2d(out_planes // 2)
        self.conv2 = conv3x3(out_planes // 2, out_planes // 4)
        self.bn3 = nn.BatchNorm2d(out_planes // 4)
        self.conv3 = conv3x3(out_planes // 4, out_planes // 4)
        self.downsample = None
        if in_planes != out_planes:
            self.downsample = nn.Sequential(nn.BatchNorm2d(in_planes), nn.ReLU(True), nn.Conv2d(in_planes, out_planes, 1, 1, bias=False))


class HourGlass(nn.Module):
    def __init__(self, num_modules, depth, num_features, first_one=False):
        super().__init__()
        self.num_modules, self.depth, self.features = num_modules, depth, num_features
        self.coordconv = CoordConvTh(64, 64, True, True, 256, first_one, out_channels=256, kernel_size=1, stride=1, padding=0)
        self._generate_network(depth)

    def _generate_network(self, level):
        self.add_module('b1_' + str(level), ConvBlock(256, 256))
        self.add_module('b2_' + str(level), ConvBlock(256, 256))
        if level > 1:
            self._generate_network(level - 1)
        else:
            self.add_module('b2_plus_' + str(level), ConvBlock(256, 256))
        self.add_module('b3_' + str(level), ConvBlock(256, 256))


def _pad_to(t, shape):
    out = torch.zeros(shape, dtype=t.dtype, device=t.device)
    out[tuple(slice(0, s) for s in t.shape)] = t
    return out


def _bn_coef(bn, pad_to=None, extra_shift=None):
    """eval-mode BatchNorm as (scale, shift) [4,C] (rows 2,3 unused); optional conv bias folded in, optional zero padding."""
    scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
    shift = bn.bias.detach() - bn.running_mean * scale
    if extra_shift is not None:
        shift = shift + extra_shift * scale
    c = torch.stack([scale, shift, scale, scale]).float()
    if pad_to is not None and pad_to > c.shape[1]:
        c = _pad_to(c, (4, pad_to))
    return c.contiguous()


class FAN(nn.Module):
    def __init__(self, num_modules=1, end_relu=False, num_landmarks=98, fname_pretrained=None, *, precision="fp32"):
        super().__init__()
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision: 'fp32' (split-bf16 MFMA products, f32 activations: the reference's arithmetic to ~1e-5) or "
                             "'bf16' (bf16 activations, 3x faster, ~1e-2)")
        self.precision = precision                    # extra keyword: the reference FAN is fp32 only
        self.num_modules, self.end_relu = num_modules, end_relu
        self.conv1 = CoordConvTh(256, 256, True, False, in_channels=3, out_channels=64, kernel_size=7, stride=2, padding=3)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = ConvBlock(64, 128)
        self.conv3 = ConvBlock(128, 128)
        self.conv4 = ConvBlock(128, 256)
        self.add_module('m0', HourGlass(1, 4, 256, first_one=True))
        self.add_module('top_m_0', ConvBlock(256, 256))
        self.add_module('conv_last0', nn.Conv2d(256, 256, 1, 1, 0))
        self.add_module('bn_end0', nn.BatchNorm2d(256))
        self.add_module('l0', nn.Conv2d(256, num_landmarks + 1, 1, 1, 0))
        self._cache = None
        self.last_raw = None
        if fname_pretrained is not None:
            self.load_pretrained_weights(fname_pretrained)

    def load_pretrained_weights(self, fname):
        checkpoint = torch.load(fname, map_location="cpu")
        model_weights = self.state_dict()
        model_weights.update({k: v for k, v in checkpoint['state_dict'].items() if k in model_weights})
        self.load_state_dict(model_weights)
        self._cache = None

    # ------------------------------------------------------------------ cached kernel-side constants (frozen, eval)
    def _block_consts(self, blk):
        n1, n2, n3 = blk.conv1.out_channels, blk.conv2.out_channels, blk.conv3.out_channels
        p2, p3 = max(n2, 64), max(n3, 64)
        c = {"n": (n1, n2, n3), "p": (p2, p3),
             "bn1": _bn_coef(blk.bn1), "bn2": _bn_coef(blk.bn2), "bn3": _bn_coef(blk.bn3, pad_to=p2),
             "w1": co.weight_layout(blk.conv1.weight.detach().float(), 0),
             "w2": co.weight_layout(_pad_to(blk.conv2.weight.detach().float(), (p2, n1, 3, 3)), 0),
             "w3": co.weight_layout(_pad_to(blk.conv3.weight.detach().float(), (p3, p2, 3, 3)), 0)}
        if blk.downsample is not None:
            c["bnd"] = _bn_coef(blk.downsample[0])
            c["wd"] = co.weight_layout(blk.downsample[2].weight.detach().float(), 0)
        return c

    def _build_cache(self, dev):
        L = _lib.lib()
        cache = {"dev": dev}
        w1 = self.conv1.conv.weight.detach().float().contiguous()                     # [64,6,7,7]
        wst = torch.empty((64, 48, 8), dtype=torch.bfloat16, device=dev)
        check(L.ppv_stem_weight_layout(ptr(w1), ptr(wst), 2, stream_ptr()), "ppv_stem_weight_layout")
        cache["stem_w"] = wst
        cache["stem_bn"] = _bn_coef(self.bn1, extra_shift=self.conv1.conv.bias.detach())
        cache["coords256"] = _coord_channels(256, 256).to(dev).contiguous()
        for name in ("conv2", "conv3", "conv4", "top_m_0"):
            cache[name] = self._block_consts(getattr(self, name))
        for name, m in self.m0.named_children():
            if isinstance(m, ConvBlock):
                cache["m0." + name] = self._block_consts(m)
        # hourglass CoordConv 1x1 (259 -> 256): conv over the 256 feature channels on MFMA + a per-pixel constant map
        # (the three coordinate channels are input independent) + bias, added by one broadcast element-wise pass
        wc = self.m0.coordconv.conv.weight.detach().float()                            # [256,259,1,1]
        cache["cc_w"] = co.weight_layout(wc[:, :256].contiguous(), 0)
        cmap = _coord_map(wc[:, 256:, 0, 0], _coord_channels(64, 64).to(dev))
        cache["cc_map"] = cmap.to(torch.bfloat16).contiguous()                        # [64,64,256]
        one = torch.ones(256, device=dev)
        cache["coords64_tail"] = _coord_channels(64, 64)[1:].to(dev).contiguous()
        cache["cc_coef"] = torch.stack([one, self.m0.coordconv.conv.bias.detach().float(), one, one]).contiguous()
        cache["last_w"] = co.weight_layout(self.conv_last0.weight.detach().float(), 0)
        cache["end_bn"] = _bn_coef(self.bn_end0, extra_shift=self.conv_last0.bias.detach())
        nl = self.l0.out_channels
        cache["l0_w"] = co.weight_layout(_pad_to(self.l0.weight.detach().float(), (128, 256, 1, 1)), 0)
        cache["l0_b"] = self.l0.bias.detach().float().contiguous()
        cache["nl"] = nl
        return cache

    # ------------------------------------------------------------------ fp32-accurate path (precision == "fp32")
    # Activations stay f32 (NHWC); every convolution runs on the bf16 MFMA kernel over a [hi | lo | hi] channel split of its
    # input against [W_hi | W_hi | W_lo] weights (csrc/fan.hip bn_act_split3_kernel), BN + ReLU fused into the split.
    @staticmethod
    def _w3(w, cout_pad=None):
        """torch conv weight f32 [Cout,Cin,R,S] -> bf16 GEMM rows over the 3-way split input (Cin3 padded to 64)."""
        w = w.detach().float()
        hi = w.bfloat16().float()
        lo = (w - hi).bfloat16().float()
        w3 = torch.cat([hi, hi, lo], dim=1)
        cin3 = (w3.shape[1] + 63) // 64 * 64
        cout = cout_pad or (w.shape[0] + 63) // 64 * 64
        return co.weight_layout(_pad_to(w3, (cout, cin3, w.shape[2], w.shape[3])), 0)

    def _block_consts_p(self, blk):
        c = {"n": (blk.conv1.out_channels, blk.conv2.out_channels, blk.conv3.out_channels),
             "bn1": _bn_coef(blk.bn1), "bn2": _bn_coef(blk.bn2), "bn3": _bn_coef(blk.bn3),
             "w1": self._w3(blk.conv1.weight), "w2": self._w3(blk.conv2.weight), "w3": self._w3(blk.conv3.weight)}
        if blk.downsample is not None:
            c["bnd"] = _bn_coef(blk.downsample[0])
            c["wd"] = self._w3(blk.downsample[2].weight)
        return c

    def _build_cache_p(self, dev):
        cache = {"dev": dev, "precise": True}
        w1 = self.conv1.conv.weight.detach().float()                                     # [64,6,7,7]: input padded to 8 channels so that
        cache["stem_w"] = self._w3(_pad_to(w1, (64, 8, 7, 7)))                           # the split runs vectorised -> [64,7,7,64]
        cache["stem_bn"] = _bn_coef(self.bn1, extra_shift=self.conv1.conv.bias.detach())
        cache["coords256"] = _coord_channels(256, 256).to(dev).contiguous()
        for name in ("conv2", "conv3", "conv4", "top_m_0"):
            cache[name] = self._block_consts_p(getattr(self, name))
        for name, m in self.m0.named_children():
            if isinstance(m, ConvBlock):
                cache["m0." + name] = self._block_consts_p(m)
        wc = self.m0.coordconv.conv.weight.detach().float()                            # [256,259,1,1]
        cache["cc_w"] = self._w3(wc[:, :256].contiguous())
        cache["cc_map"] = (_coord_map(wc[:, 256:, 0, 0], _coord_channels(64, 64).to(dev))
                           + self.m0.coordconv.conv.bias.detach().float()).contiguous()   # [64,64,256] f32, bias folded in
        cache["last_w"] = self._w3(self.conv_last0.weight)
        cache["end_bn"] = _bn_coef(self.bn_end0, extra_shift=self.conv_last0.bias.detach())
        cache["end_bn_id"] = None
        cache["l0_w"] = self._w3(self.l0.weight, cout_pad=128)
        cache["l0_b"] = self.l0.bias.detach().float().contiguous()
        cache["nl"] = self.l0.out_channels
        return cache

    @staticmethod
    def _split3(x, coef, relu, C=None):
        """x [B,H,W,ld] f32 (first C channels used) -> act(x*scale+shift) as bf16 [B,H,W,pad64(3C)] = [hi | lo | hi]."""
        B, H, W, ld = x.shape
        C = C or ld
        cp = (3 * C + 63) // 64 * 64
        y = torch.empty((B, H, W, cp), dtype=torch.bfloat16, device=x.device)
        check(_lib.lib().ppv_bn_act_split3(ptr(x), ptr(coef), ptr(y), B * H * W, C, cp, int(relu), ld, stream_ptr()), "ppv_bn_act_split3")
        return y

    def _conv_p(self, x, coef, w3, stride=1, pad=0, relu=True, C=None):
        """(BN + ReLU +) convolution, f32 in / f32 out; the output keeps the GEMM's 64-padded channel count (callers pass the
        real count on as ``C`` / a source stride instead of slicing)."""
        return co.conv_fwd(self._split3(x, coef, relu, C), w3, stride, pad, out_f32=True)

    def _convblock_p(self, x, c):
        n1, n2, n3 = c["n"]
        o1 = self._conv_p(x, c["bn1"], c["w1"], 1, 1)
        o2 = self._conv_p(o1, c["bn2"], c["w2"], 1, 1, C=n1)
        o3 = self._conv_p(o2, c["bn3"], c["w3"], 1, 1, C=n2)
        res = x if "wd" not in c else self._conv_p(x, c["bnd"], c["wd"], 1, 0)
        B, H, W, _ = x.shape
        out = torch.empty((B, H, W, n1 + n2 + n3), dtype=torch.float32, device=x.device)
        check(_lib.lib().ppv_concat3_add(ptr(o1), ptr(o2), ptr(o3), ptr(res), ptr(out), B * H * W, n1, n2, n3, o1.shape[-1],
                                         o2.shape[-1], o3.shape[-1], 1, stream_ptr()), "ppv_concat3_add")
        return out

    @staticmethod
    def _avgpool_p(x):
        B, H, W, C = x.shape
        y = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
        check(_lib.lib().ppv_avgpool2_nhwc(ptr(x), ptr(y), B, H, W, C, 1, stream_ptr()), "ppv_avgpool2_nhwc")
        return y

    def _hourglass_p(self, level, x, cache):
        up1 = self._convblock_p(x, cache[f"m0.b1_{level}"])
        low = self._convblock_p(self._avgpool_p(x), cache[f"m0.b2_{level}"])
        low = self._hourglass_p(level - 1, low, cache) if level > 1 else self._convblock_p(low, cache["m0.b2_plus_1"])
        low = self._convblock_p(low, cache[f"m0.b3_{level}"])
        B, H, W, C = up1.shape
        out = torch.empty_like(up1)
        check(_lib.lib().ppv_upsample2_add(ptr(up1), ptr(low), ptr(out), B, H, W, C, 1, stream_ptr()), "ppv_upsample2_add")   # nearest x2 (wing.py:69)
        return out

    def _trunk_p(self, x6):
        cache = self._cache
        x = torch.nn.functional.pad(x6.permute(0, 2, 3, 1), (0, 2)).contiguous()         # [B,256,256,8] f32 (two zero channels)
        x = self._conv_p(x, None, cache["stem_w"], 2, 3, relu=False)                      # CoordConv 7x7/2 (bias folded into bn)
        x = torch.relu(x * cache["stem_bn"][0] + cache["stem_bn"][1])
        x = self._avgpool_p(self._convblock_p(x, cache["conv2"]))
        x = self._convblock_p(self._convblock_p(x, cache["conv3"]), cache["conv4"])
        h = self._conv_p(x, None, cache["cc_w"], 1, 0, relu=False) + cache["cc_map"]
        ll = self._convblock_p(self._hourglass_p(4, h, cache), cache["top_m_0"])
        ll = self._conv_p(ll, cache["end_bn_id"], cache["last_w"], 1, 0, relu=False)
        return self._conv_p(ll, cache["end_bn"], cache["l0_w"], 1, 0, relu=True)          # BN + ReLU fused into the head conv's split

    def refresh(self):
        """Call after changing parameters in place (the kernel-side constants are cached)."""
        self._cache = None

    # ------------------------------------------------------------------ forward pieces
    def _convblock(self, x, c):
        n1, n2, n3 = c["n"]
        p2, p3 = c["p"]
        o1 = co.conv_fwd(co.bn_act(x, c["bn1"]), c["w1"], 1, 1)
        o2 = co.conv_fwd(co.bn_act(o1, c["bn2"]), c["w2"], 1, 1)
        o3 = co.conv_fwd(co.bn_act(o2, c["bn3"]), c["w3"], 1, 1)
        res = x if "wd" not in c else co.conv_fwd(co.bn_act(x, c["bnd"]), c["wd"], 1, 0)
        B, H, W, _ = x.shape
        out = torch.empty((B, H, W, n1 + n2 + n3), dtype=torch.bfloat16, device=x.device)
        check(_lib.lib().ppv_concat3_add(ptr(o1), ptr(o2), ptr(o3), ptr(res), ptr(out), B * H * W, n1, n2, n3, n1, p2, p3, 0,
                                         stream_ptr()), "ppv_concat3_add")
        return out

    def _avgpool(self, x):
        B, H, W, C = x.shape
        y = torch.empty((B, H // 2, W // 2, C), dtype=torch.bfloat16, device=x.device)
        check(_lib.lib().ppv_avgpool2_nhwc(ptr(x), ptr(y), B, H, W, C, 0, stream_ptr()), "ppv_avgpool2_nhwc")
        return y

    def _hourglass(self, level, x, cache):
        up1 = self._convblock(x, cache[f"m0.b1_{level}"])
        low = self._convblock(self._avgpool(x), cache[f"m0.b2_{level}"])
        low = self._hourglass(level - 1, low, cache) if level > 1 else self._convblock(low, cache["m0.b2_plus_1"])
        low = self._convblock(low, cache[f"m0.b3_{level}"])
        B, H, W, C = up1.shape
        out = torch.empty_like(up1)
        check(_lib.lib().ppv_upsample2_add(ptr(up1), ptr(low), ptr(out), B, H, W, C, 0, stream_ptr()), "ppv_upsample2_add")
        return out

    def _trunk(self, x6):
        """x6 [B,6,256,256] f32 NCHW (image*0.5+0.5 and the three coordinate channels) -> l0 raw [B,64,64,128] f32."""
        if self.precision == "fp32":
            return self._trunk_p(x6)
        cache = self._cache
        L = _lib.lib()
        B = x6.shape[0]
        raw0 = torch.empty((B, 128, 128, 64), dtype=torch.bfloat16, device=x6.device)
        check(L.ppv_stem_conv6(ptr(x6), ptr(cache["stem_w"]), ptr(raw0), B, 256, 256, stream_ptr()), "ppv_stem_conv6")
        x = co.bn_act(raw0, cache["stem_bn"])
        x = self._avgpool(self._convblock(x, cache["conv2"]))
        x = self._convblock(self._convblock(x, cache["conv3"]), cache["conv4"])
        h = co.conv_fwd(x, cache["cc_w"], 1, 0)
        h = co.bn_act(h, cache["cc_coef"], res=cache["cc_map"], relu=False, res_broadcast=True)
        ll = self._convblock(self._hourglass(4, h, cache), cache["top_m_0"])
        ll = co.bn_act(co.conv_fwd(ll, cache["last_w"], 1, 0), cache["end_bn"])
        return co.conv_fwd(ll, cache["l0_w"], 1, 0, out_f32=True)       # head logits stay in fp32 (49-channel sums follow)

    def _prepare(self, x):
        if self.training:
            raise NotImplementedError("ppv_amd FAN implements the eval-mode forward the reference uses (model.py:298-306)")
        if not x.is_cuda:
            raise RuntimeError("ppv_amd FAN runs on an MI355X (input must be a cuda tensor); no CPU path")
        precise = self.precision == "fp32"
        if self._cache is None or self._cache["dev"] != x.device or self._cache.get("precise", False) != precise:
            self._cache = self._build_cache_p(x.device) if precise else self._build_cache(x.device)

    @torch.no_grad()
    def get_heatmap(self, x, b_preprocess=True, Privacy=False, delimiter=False):
        """wing.py:240-260.  Only the ``Privacy=True`` branch (solver.py:147) runs on the device; the other branches
        post-process landmarks on the host with OpenCV (``preprocess``, wing.py:440-578: out of scope)."""
        if not (b_preprocess and Privacy):
            raise NotImplementedError("host-side landmark post-processing (wing.py preprocess) is out of scope")
        self._prepare(x)
        L = _lib.lib()
        x = x.detach().float().contiguous()
        B, _, Hin, Win = x.shape
        x6 = torch.empty((B, 6, 256, 256), dtype=torch.float32, device=x.device)
        check(L.ppv_fan_input(ptr(x), ptr(self._cache["coords256"]), ptr(x6), B, Hin, Win, 256, stream_ptr()), "ppv_fan_input")
        raw = self._trunk(x6)
        nl = self._cache["nl"]
        self.last_raw = torch.empty((B, nl, 64, 64), dtype=torch.float32, device=x.device)
        sums = torch.empty((B, 2, 64, 64), dtype=torch.float32, device=x.device)
        heat = torch.empty((B, 2, 256, 256), dtype=torch.float32, device=x.device)
        check(L.ppv_fan_head(ptr(raw), ptr(self._cache["l0_b"]), ptr(self.last_raw), ptr(sums), ptr(heat), B, 64, 128, nl, 49,
                             nl - 1, 4, stream_ptr()), "ppv_fan_head")
        return [heat[:, 0:1].clamp_(0, 1), heat[:, 1:2].clamp_(0, 1)]

    # ------------------------------------------------------------------ differentiable path (wing.py:262-272 get_heatmap_train)
    # The landmark loss of the de-identification training back-propagates THROUGH the frozen regressor into the generated image.
    # Same arithmetic as the fp32-accurate forward (split-bf16 MFMA products, f32 activations), expressed with
    # ppv_amd.nn_ops.conv2d_f32 (forward + data gradient on the MFMA kernels) and f32 torch element-wise ops for the eval-mode
    # BatchNorm affine, ReLU, concatenation, pooling and resampling, so autograd carries the gradient to ``x``.  The regressor's
    # own parameters are frozen in the reference (eval mode, no optimiser holds them); their gradients are not produced here.
    @staticmethod
    def _bn_relu_t(bn, t):
        scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
        return torch.relu(t * scale + (bn.bias.detach() - bn.running_mean * scale))

    def _convblock_t(self, blk, x):
        from .nn_ops import conv2d_f32
        o1 = conv2d_f32(self._bn_relu_t(blk.bn1, x), blk.conv1.weight, None, 1, 1, weight_grad=False)
        o2 = conv2d_f32(self._bn_relu_t(blk.bn2, o1), blk.conv2.weight, None, 1, 1, weight_grad=False)
        o3 = conv2d_f32(self._bn_relu_t(blk.bn3, o2), blk.conv3.weight, None, 1, 1, weight_grad=False)
        res = x if blk.downsample is None else conv2d_f32(self._bn_relu_t(blk.downsample[0], x), blk.downsample[2].weight, None, 1, 0, weight_grad=False)
        return torch.cat((o1, o2, o3), dim=-1) + res

    def _hourglass_t(self, level, x):
        m = self.m0._modules
        up1 = self._convblock_t(m[f"b1_{level}"], x)
        B, H, W, C = x.shape
        low = self._convblock_t(m[f"b2_{level}"], x.view(B, H // 2, 2, W // 2, 2, C).mean(dim=(2, 4)))
        low = self._hourglass_t(level - 1, low) if level > 1 else self._convblock_t(m["b2_plus_1"], low)
        low = self._convblock_t(m[f"b3_{level}"], low)
        return up1 + low.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)          # nearest x2 (wing.py:69)

    def forward_train(self, x):
        """x [B,3,256,256] in [0,1], autograd enabled -> ([heat-map logits [B,99,64,64]], [boundary channel])  (wing.py:212-238)."""
        from .nn_ops import conv2d_f32
        if self.training:
            raise NotImplementedError("ppv_amd FAN is the frozen (eval-mode BatchNorm) regressor of the reference (model.py:298-306)")
        if not x.is_cuda:
            raise RuntimeError("ppv_amd FAN runs on an MI355X (input must be a cuda tensor); no CPU path")
        if x.shape[-1] != 256 or x.shape[-2] != 256:
            raise RuntimeError("FAN's CoordConv is built for 256 x 256 inputs (wing.py:184)")
        B = x.shape[0]
        dev = x.device
        c256 = _coord_channels(256, 256).to(dev).permute(1, 2, 0)                        # [256,256,3]
        c64 = _coord_channels(64, 64).to(dev).permute(1, 2, 0)
        t = torch.cat([x.float().permute(0, 2, 3, 1), c256.unsqueeze(0).expand(B, -1, -1, -1)], dim=-1)      # CoordConv: + xx, yy, rr
        t = conv2d_f32(t, self.conv1.conv.weight, self.conv1.conv.bias, 2, 3, weight_grad=False)
        t = self._bn_relu_t(self.bn1, t)
        t = self._convblock_t(self.conv2, t)
        t = t.view(B, 64, 2, 64, 2, t.shape[-1]).mean(dim=(2, 4))                          # F.avg_pool2d(., 2)
        t = self._convblock_t(self.conv4, self._convblock_t(self.conv3, t))
        cc = self.m0.coordconv.conv
        h = conv2d_f32(torch.cat([t, c64.unsqueeze(0).expand(B, -1, -1, -1)], dim=-1), cc.weight, cc.bias, 1, 0, weight_grad=False)
        ll = self._convblock_t(self.top_m_0, self._hourglass_t(4, h))
        ll = conv2d_f32(ll, self.conv_last0.weight, self.conv_last0.bias, 1, 0, weight_grad=False)
        ll = self._bn_relu_t(self.bn_end0, ll)
        out = conv2d_f32(ll, self.l0.weight, self.l0.bias, 1, 0, weight_grad=False).permute(0, 3, 1, 2)
        if self.end_relu:
            out = F.relu(out)
        boundary = c64.permute(2, 0, 1)[1:3].unsqueeze(0).expand(B, -1, -1, -1)             # last two CoordConv input channels
        return [out], [boundary]

    def get_heatmap_train(self, x, b_preprocess=True, Privacy=False, delimiter=False):
        """wing.py:262-272: 0-1 normalised heat-maps WITH autograd (gradient w.r.t. ``x``)."""
        from .nn_ops import bilinear_resize
        x = bilinear_resize(x, size=256)                                    # F.interpolate(x, size=256, mode='bilinear'), wing.py:264
        outputs, _ = self.forward_train(x * 0.5 + 0.5)
        heatmaps = outputs[-1][:, :-1, :, :]
        scale_factor = x.size(2) // heatmaps.size(2)
        if b_preprocess and Privacy:
            heatmaps = bilinear_resize(heatmaps, scale_factor=scale_factor, align_corners=True)       # wing.py:270
            heatmaps = [heatmaps[:, :49].sum(dim=1, keepdim=True).clamp_(0, 1), heatmaps[:, 49:].sum(dim=1, keepdim=True).clamp_(0, 1)]
        return heatmaps

    @torch.no_grad()
    def forward(self, x):
        """x [B,3,256,256] in [0,1] -> ([heat-map logits [B,99,64,64] f32], [boundary channel])  (wing.py:212-238)."""
        self._prepare(x)
        L = _lib.lib()
        x = x.detach().float().contiguous()
        B = x.shape[0]
        if x.shape[-1] != 256 or x.shape[-2] != 256:
            raise RuntimeError("FAN's CoordConv is built for 256 x 256 inputs (wing.py:184)")
        coords = self._cache["coords256"].unsqueeze(0).expand(B, -1, -1, -1)
        x6 = torch.cat([x, coords], 1).contiguous()
        raw = self._trunk(x6)
        nl = self._cache["nl"]
        out = torch.empty((B, nl, 64, 64), dtype=torch.float32, device=x.device)
        sums = torch.empty((B, 2, 64, 64), dtype=torch.float32, device=x.device)
        heat = torch.empty((B, 2, 256, 256), dtype=torch.float32, device=x.device)
        check(L.ppv_fan_head(ptr(raw), ptr(self._cache["l0_b"]), ptr(out), ptr(sums), ptr(heat), B, 64, 128, nl, 49, nl - 1, 4,
                             stream_ptr()), "ppv_fan_head")
        if self.end_relu:
            out = F.relu(out)
        boundary = self._cache["coords64_tail"].unsqueeze(0).expand(B, -1, -1, -1)      # last two CoordConv input channels
        return [out], [boundary]
