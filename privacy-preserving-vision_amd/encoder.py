"""Drop-in for reference ``Image_Caption/models.py:8 Encoder`` (torchvision ResNet-101 trunk) on MI355X.

Same constructor, ``forward(images) -> [B, E, E, 2048]``, ``fine_tune()``, attribute ``.resnet`` with
torchvision-compatible ``state_dict`` keys (``resnet.0.weight``, ``resnet.1.*``, ``resnet.4..7.<i>.conv1.weight`` ...)
and ``.adaptive_pool``.  The ``torch.nn`` sub-modules are PARAMETER HOLDERS only: the whole trunk runs as one
``torch.autograd.Function`` whose forward/backward launch the hand-written HIP kernels of libppv_hip.so on NHWC
bfloat16 activations (fp32 accumulation, fp32 BatchNorm statistics, fp32 weight gradients).
"""
import torch
from torch import nn

from . import convops as co
from . import _lib

LAYERS = (3, 4, 23, 3)
PLANES = (64, 128, 256, 512)


class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck parameter layout (v1.5: stride on conv2)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("parameter holder: the trunk runs through ppv_amd.encoder.Encoder.forward")


def _make_trunk(layers):
    inplanes = 64
    mods = [nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, stride=2, padding=1)]
    for i, (planes, n) in enumerate(zip(PLANES, layers)):
        stride = 1 if i == 0 else 2
        blocks = []
        for b in range(n):
            ds = None
            if b == 0 and (stride != 1 or inplanes != planes * 4):
                ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
            blocks.append(Bottleneck(inplanes, planes, stride if b == 0 else 1, ds))
            inplanes = planes * 4
        mods.append(nn.Sequential(*blocks))
    net = nn.Sequential(*mods)
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return net


class _ConvRec:
    """one conv + its BatchNorm, with cached bf16 kernel layouts keyed on the parameter version."""
    __slots__ = ("conv", "bn", "k", "stride", "pad", "_wt", "_wd", "_ver_t", "_ver_d", "stem", "wl", "wl_idx")

    def __init__(self, conv, bn, stem=False):
        self.conv, self.bn = conv, bn
        self.k, self.stride, self.pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        self._wt = self._wd = None
        self._ver_t = self._ver_d = -1
        self.stem = stem
        self.wl, self.wl_idx = None, -1          # shared WeightLayouts of the trainable convs (one refresh launch per step)

    # Frozen weights are converted once; trainable ones every step (fused optimisers update ``.data`` without a
    # reliable version bump, so the cache key is (version, step token)).
    def wt(self, token=0):
        w = self.conv.weight
        if self.wl is not None and w.requires_grad:
            return self.wl.fwd[self.wl_idx]
        if w.requires_grad:
            if self._wt is None or self._ver_t != token:
                self._wt = co.weight_layout(w.detach(), 0)
                self._ver_t = token
            return self._wt
        key = (w._version, w.data_ptr())             # `.data = ...` swaps the storage without a version bump
        if self._wt is None or self._ver_t != key or self._wt.device != w.device:
            self._wt = co.stem_weight_layout(w.detach(), 0) if self.stem else co.weight_layout(w.detach(), 0)
            self._ver_t = key
        return self._wt

    def wd(self, token=0):
        w = self.conv.weight
        if self.wl is not None and w.requires_grad:
            return self.wl.dg[self.wl_idx]
        if w.requires_grad:
            if self._wd is None or self._ver_d != token:
                self._wd = co.weight_layout(w.detach(), 1)
                self._ver_d = token
            return self._wd
        key = (w._version, w.data_ptr())
        if self._wd is None or self._ver_d != key or self._wd.device != w.device:
            self._wd = co.stem_weight_layout(w.detach(), 1) if self.stem else co.weight_layout(w.detach(), 1)
            self._ver_d = key
        return self._wd

    def invalidate(self):
        self._wt = self._wd = None


def _bn_momentum(bn):
    # momentum=None is torch's cumulative moving average: factor 1 / (batches seen, this one included)
    return bn.momentum if bn.momentum is not None else 1.0 / (int(bn.num_batches_tracked) + 1)


def _bn_coef(rec, stat_part, count):
    bn = rec.bn
    if bn.training:
        return co.bn_finalize(stat_part, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                              _bn_momentum(bn), bn.eps)
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    scale = bn.weight.detach() * invstd
    return torch.stack([scale, bn.bias.detach() - bn.running_mean * scale, bn.running_mean, invstd]).contiguous()


_META_GETTERS = frozenset(("shape", "dtype", "device", "requires_grad", "is_cuda", "ndim", "grad_fn", "is_leaf", "layout", "names", "is_sparse",
                           "is_quantized", "is_meta", "is_cpu", "output_nr"))
_META_METHODS = frozenset(("size", "dim", "numel", "stride", "is_contiguous", "element_size", "nelement", "ndimension", "is_floating_point",
                           "is_complex", "get_device", "storage_offset", "__len__", "__hash__", "dim_order", "is_pinned", "is_shared"))


class LazyEncoderOut(torch.Tensor):
    """What ``Encoder.forward`` returns with ``lazy_output`` on: the [B,E,E,2048] float32 tensor of models.py:39-41 whose VALUES are
    written the first time anything reads them.

    The reference's ``AdaptiveAvgPool2d(36)`` on an 8 x 8 map + permute replicates every cell into a 4-5 x 4-5 window: 1.36 GB of f32 at
    B = 128 that ppv_amd's own consumers never read (``ppv_amd.decoder`` and the benchmark's head work on the 8 x 8 cell map behind it,
    attribute ``_ppv_cells``).  The buffer is allocated in forward, its shape / dtype / device / ``grad_fn`` are those of the real
    output, and the pooling kernel (csrc/trunk_ops.hip adaptive_pool_fwd) runs on the first torch operation that takes this tensor as
    an argument (``__torch_function__``: arithmetic, indexing, ``.cpu()``, ``data_ptr()``, ``print`` ...), which then sees the ordinary
    autograd-connected tensor.  Not covered: the functional autograd entry points called ON this object itself
    (``torch.autograd.grad(lazy, ...)`` / ``torch.autograd.backward([lazy])``) and ``torch.autograd.Function.apply(lazy)`` do not dispatch
    through ``__torch_function__``: the object stays attached to the graph (gradients flow), and its values are written as soon as
    the callee's first torch operation touches it; code that reads the storage without any torch call must ``.materialize()``
    first, or construct the encoder with ``Encoder(lazy_output=False)``."""

    @classmethod
    def wrap(cls, real, fill):
        """real: the trunk's (not yet written) output tensor; fill(): launches the kernel that writes it."""
        with torch._C.DisableTorchFunctionSubclass():
            t = real.as_subclass(cls)          # same storage, stays attached to the autograd graph (raw autograd consumers see the edge)
        t._real, t._fill = real, fill
        return t

    def materialize(self):
        """Write the values (once) and return the ordinary tensor."""
        fill = self.__dict__.get("_fill")
        if fill is not None:
            self.__dict__["_fill"] = None
            fill()
        return self.__dict__["_real"]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        owner = getattr(func, "__self__", None)
        name = getattr(owner, "__name__", None) if getattr(func, "__name__", "") == "__get__" else getattr(func, "__name__", None)
        meta = (getattr(func, "__name__", "") == "__get__" and name in _META_GETTERS) or (name in _META_METHODS and owner is None)

        def sub(a):
            if isinstance(a, LazyEncoderOut):
                return a.__dict__["_real"] if meta else a.materialize()
            if isinstance(a, (tuple, list)):
                return type(a)(sub(v) for v in a)
            return a

        with torch._C.DisableTorchFunctionSubclass():
            return func(*[sub(a) for a in args], **{k: sub(v) for k, v in kwargs.items()})


class _BlockSave:
    """What the forward of one bottleneck keeps for backward, for the blocks that go through ppv_bottleneck_fwd: the block input and
    output as tensors, everything in between (raw conv outputs, activations, the output's sign mask, the three BatchNorm coefficient
    sets) in ONE byte arena -- one ``torch.empty`` instead of nine.  Reads like the 13-tuple the per-kernel path stores
    (xin, x1, c1, y1, x2, c2, y2, x3, c3, xd, cd, yout, xin_bits): indexing builds the view on demand (the per-kernel backward, the
    tests' taps); ppv_bottleneck_bwd takes the addresses from ``ptr``."""
    __slots__ = ("xin", "yout", "arena", "off", "shp", "prev", "ptr_bits", "prev_ptr")
    _BF, _F32, _U8 = torch.bfloat16, torch.float32, torch.uint8

    def __init__(self, xin, yout, arena, P, prev):
        Bn, H, W, C3 = xin.shape
        M = Bn * H * W
        a256 = lambda n: (n + 255) & ~255
        t, T, cc = a256(M * P * 2), a256(M * C3 * 2), a256(4 * P * 4)
        # x1, y1, x2, y2, x3, bits, c1, c2, c3
        o = [0, t, 2 * t, 3 * t, 4 * t, 4 * t + T]
        o.append(o[-1] + a256(M * C3 // 8)); o.append(o[-1] + cc); o.append(o[-1] + cc)
        self.xin, self.yout, self.arena, self.off, self.prev = xin, yout, arena, o, prev
        self.shp = (Bn, H, W, P)
        self.ptr_bits = arena.data_ptr() + o[5] if arena is not None else None
        # address of the block INPUT's sign mask: the previous block's (a thunk when that block is a _BlockSave), none for the first
        self.prev_ptr = None if prev is None else (prev.__self__.ptr_bits if callable(prev) else prev.data_ptr())

    @staticmethod
    def nbytes(M, P):
        a256 = lambda n: (n + 255) & ~255
        return 4 * a256(M * P * 2) + a256(M * 4 * P * 2) + a256(M * 4 * P // 8) + 2 * a256(16 * P) + a256(64 * P)

    def ptr(self, k):                                   # k: 0 x1, 1 y1, 2 x2, 3 y2, 4 x3, 5 bits, 6 c1, 7 c2, 8 c3
        return self.arena.data_ptr() + self.off[k]

    def _view(self, k):
        Bn, H, W, P = self.shp
        M = Bn * H * W
        if k <= 3:
            return self.arena[self.off[k]: self.off[k] + M * P * 2].view(self._BF).view(Bn, H, W, P)
        if k == 4:
            return self.arena[self.off[4]: self.off[4] + M * 8 * P].view(self._BF).view(Bn, H, W, 4 * P)
        if k == 5:
            return self.arena[self.off[5]: self.off[5] + M * 4 * P // 8]
        C = 4 * P if k == 8 else P
        return self.arena[self.off[k]: self.off[k] + 16 * C].view(self._F32).view(4, C)

    def bits(self):
        return self._view(5)

    def __len__(self):
        return 13

    def __getitem__(self, i):
        if i < 0:
            i += 13
        if i == 0:
            return self.xin
        if i == 11:
            return self.yout
        if i in (9, 10):
            return None
        if i == 12:
            return self.prev() if callable(self.prev) else self.prev
        return self._view({1: 0, 2: 6, 3: 1, 4: 2, 5: 7, 6: 3, 7: 4, 8: 8}[i])

    def __iter__(self):
        return (self[i] for i in range(13))


def masked_stream(dev, first_cu, n_cus):
    """A stream whose kernels run on CUs [first_cu, first_cu + n_cus) only (csrc/trunk_plan.hip ppv_stream_create_masked), as a torch
    stream object.  The handle lives as long as the process."""
    h = _lib.ctypes.c_void_p()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().ppv_stream_create_masked(_lib.ctypes.byref(h), int(first_cu), int(n_cus)), "ppv_stream_create_masked")
    return torch.cuda.ExternalStream(h.value, device=dev)


def _make_side_stream(dev):
    """The stream the weight gradients run on beside the data-gradient / BatchNorm chain.  PPV_WGRAD_CUS=n (A/B): restricted
    to n of the 256 CUs (spread over the XCDs); PPV_BWD_MAIN_CUS=m then runs the rest of the trunk's backward on the OTHER m CUs."""
    import os
    n = int(os.environ.get("PPV_WGRAD_CUS", "0"))
    if n > 0:
        return masked_stream(dev, 0, n)
    return torch.cuda.Stream(device=dev)


def _bwd_main_stream(enc, dev):
    """None, or the masked stream the data-gradient / BatchNorm chain of backward runs on (PPV_BWD_MAIN_CUS=m: the top m CUs)."""
    import os
    m = int(os.environ.get("PPV_BWD_MAIN_CUS", "0"))
    if m <= 0:
        return None
    st = enc.__dict__.get("_bwd_main_stream")
    if st is None or st[0] != m:
        st = (m, masked_stream(dev, 256 - m, m))
        enc.__dict__["_bwd_main_stream"] = st
    return st[1]


class _TrunkFn(torch.autograd.Function):
    """(images f32 NCHW, *params) -> [B,E,E,2048] f32.  params follow Encoder._param_list()."""

    @staticmethod
    def forward(ctx, enc, images, *params):
        images = images.contiguous().float()
        B, _, H, W = images.shape
        if H % 32 or W % 32:
            # stem /2, max-pool /2, three stride-2 blocks: the kernels' tile geometry (and the BatchNorm sample counts) assume
            # every stage halves exactly; torchvision would floor odd sizes (the reference only feeds 256 x 256, datasets.py:46)
            raise ValueError(f"ppv_amd Encoder: image height and width must be multiples of 32, got {H} x {W}")
        dev = images.device
        train = enc.resnet[1].training
        saved = {}
        import os as _os0

        enc._step_token += 1
        tok = enc._step_token
        if getattr(enc, "_wl_prefetched", False):    # Encoder.prefetch_weight_layouts() already converted the updated weights
            object.__setattr__(enc, "_wl_prefetched", False)
        else:
            enc._refresh_weight_layouts()
        # Default train-mode step: the whole trunk from ONE FFI call over a persistent arena (ppv_amd/trunk_exec.py, csrc/trunk_plan.hip:
        # the launches below in the same order, enqueued from C; no allocation, one memset).  The per-kernel form that follows is the
        # reference the tests' taps, the per-class timing pass, eval mode and the opt-in schedules run (PPV_TRUNK_PLAN=0 forces it).
        from . import trunk_exec as _tx
        holder = None
        blocks = []
        if _tx.usable(enc, train):
            x, holder = _tx.forward(enc, images, tok, enc._param_list())
        if holder is None:
            # one zeroed f32 pool for every conv's BN partial sums of this step ([rows<=32][2][C] each)
            pool = torch.zeros(enc._stat_pool_elems(B, H, W), dtype=torch.float32, device=dev) if train else None
            pool_off = [0]

            # Train-mode BatchNorms without a projection partner take their statistics in TWO partial rows and the apply kernel derives the
            # coefficients itself (co.bn_act_fold, csrc/trunk_ops.hip bn_act_fold_wg_kernel: thread j of every workgroup computes channel j's
            # scale / shift while the rows are in flight): no bn_finalize launch between convolution and apply pass (92 of 104 per step).
            # MEASURED (round 3, whole step, every configuration twice on one box): +0.7 .. +1.1 % (5566 -> 5619, 5497 -> 5532, 5619 -> 5683,
            # 5603 -> 5644 images/s); one row: -0.3 .. 0 % (128 - 512 row tiles adding to one address at the end of the convolution), four
            # rows: +0.4 %, eight: -1.5 %.  The first two forms of the fold lost: a 512-workgroup looped kernel (round 2, PPV_BN_FUSED) and
            # per-THREAD coefficients from one row (-1.3 %; PPV_BN_FOLD_THREAD=1 keeps it reachable).  PPV_BN_FOLD_ACT=0: bn_finalize + bn_act.
            fold_act = train and _os0.environ.get("PPV_BN_FOLD_ACT", "1") == "1"
            # PPV_BLOCK_EXEC=0: every kernel of a bottleneck through its own FFI call (the form the roofline pass and the tests' taps use)
            block_exec = train and _os0.environ.get("PPV_BLOCK_EXEC", "1") != "0"
            bargs = _lib.BottleneckFwd()
            bargs.zero_page = co.zero_page(dev).data_ptr()
            fold_rows = max(1, int(_os0.environ.get("PPV_BN_FOLD_ROWS", "2")))     # partial rows the fold path's convolutions leave (adders per address = row tiles / this)
            if torch.are_deterministic_algorithms_enabled():
                fold_rows = 32             # 32 partial rows: <= 2 adders per address up to 8192 pixels (bit-reproducible there), fewest possible beyond

            def part_for(M, C, one_row=False):
                if not train:
                    return None
                rows = min(fold_rows, co.stat_tiles(M)) if one_row else co.stat_tiles(M)
                n = rows * 2 * C
                v = pool[pool_off[0]:pool_off[0] + n].view(rows, 2, C)
                pool_off[0] += n
                return v

            # ---- stem: conv 7x7/2 + BN + ReLU + maxpool 3x3/2  (resnet.0-3)
            st = enc._stem
            Ho, Wo = H // 2, W // 2
            p0 = part_for(B * Ho * Wo, 64)
            raw0 = co.stem_conv(images, st.wt(tok), p0)
            c0 = _bn_coef(st, p0, B * Ho * Wo)
            y0, arg0 = co.bn_relu_maxpool(raw0, c0)
            saved["stem"] = (raw0, c0, y0, arg0)
            x = y0
            blocks = []
            import os as _os
            fused_bn = train and _os.environ.get("PPV_BN_FUSED", "0") == "1"      # 1: bn_finalize + bn_act in one launch (measured slower: DESIGN 4b)
            want_bits = train or bool(getattr(enc, "_grad_wanted", False))        # eval mode under autograd: backward needs the ReLU masks too
            xin_bits = None        # (block input > 0) bit mask; the first block's input is the max-pool output, masked by its own backward
            for blk in enc._blocks:
                xin = x
                r1, r2, r3, rd = blk
                Bn, Hin, Win, _ = xin.shape
                use_fold = fold_act and not fused_bn and all(r_.bn.training for r_ in blk if r_ is not None)
                if use_fold and block_exec and rd is None and r2.stride == 1 and co.PROFILE is None:
                    # one FFI crossing for the whole block (csrc/block_exec.hip: the same six launches in the same order)
                    P_ = r1.conv.out_channels
                    M1 = Bn * Hin * Win
                    T_ = min(fold_rows, co.stat_tiles(M1))
                    n1, n3 = T_ * 2 * P_, T_ * 8 * P_
                    sbase = pool.data_ptr() + 4 * pool_off[0]
                    pool_off[0] += 2 * n1 + n3
                    arena = torch.empty(_BlockSave.nbytes(M1, P_), dtype=torch.uint8, device=dev)
                    yout = torch.empty((Bn, Hin, Win, 4 * P_), dtype=torch.bfloat16, device=dev)
                    bs = _BlockSave(xin, yout, arena, P_, xin_bits)
                    ap, ao = arena.data_ptr(), bs.off
                    a = bargs
                    a.xin, a.w1, a.w2, a.w3 = xin.data_ptr(), r1.wt(tok).data_ptr(), r2.wt(tok).data_ptr(), r3.wt(tok).data_ptr()
                    a.x1, a.y1, a.x2, a.y2, a.x3, a.yout, a.bits = ap + ao[0], ap + ao[1], ap + ao[2], ap + ao[3], ap + ao[4], yout.data_ptr(), ap + ao[5]
                    a.stats1, a.stats2, a.stats3 = sbase, sbase + 4 * n1, sbase + 8 * n1
                    a.coef1, a.coef2, a.coef3 = ap + ao[6], ap + ao[7], ap + ao[8]
                    for i_, r_ in ((1, r1), (2, r2), (3, r3)):
                        bn_ = r_.bn
                        setattr(a, "g%d" % i_, bn_.weight.data_ptr()); setattr(a, "b%d" % i_, bn_.bias.data_ptr())
                        setattr(a, "rm%d" % i_, bn_.running_mean.data_ptr() if bn_.running_mean is not None else None)
                        setattr(a, "rv%d" % i_, bn_.running_var.data_ptr() if bn_.running_var is not None else None)
                        setattr(a, "mom%d" % i_, _bn_momentum(bn_)); setattr(a, "eps%d" % i_, bn_.eps)
                    a.B, a.H, a.W, a.Cin, a.planes, a.stride, a.T1, a.T2, a.T3 = Bn, Hin, Win, xin.shape[3], P_, 1, T_, T_, T_
                    _lib.check(_lib.lib().ppv_bottleneck_fwd(_lib.ctypes.byref(a), _lib.stream_ptr()), "ppv_bottleneck_fwd")
                    blocks.append(bs)
                    x, xin_bits = yout, bs.bits          # the sign mask as a thunk: a view only if the per-kernel path asks for one
                    continue
                if callable(xin_bits):
                    xin_bits = xin_bits()
                p = part_for(Bn * Hin * Win, r1.conv.out_channels, one_row=use_fold)
                x1 = co.conv_fwd(xin, r1.wt(tok), 1, 0, p)
                H2, W2 = Hin // r2.stride, Win // r2.stride
                if fused_bn and all(r_.bn.training for r_ in blk if r_ is not None):
                    # train mode: statistics -> coefficients -> apply in one launch per BatchNorm (co.bn_act_train)
                    y1, _, c1, _ = co.bn_act_train(x1, p, Bn * Hin * Win, r1.bn, _bn_momentum(r1.bn))
                    p = part_for(Bn * H2 * W2, r2.conv.out_channels)
                    x2 = co.conv_fwd(y1, r2.wt(tok), r2.stride, 1, p)
                    y2, _, c2, _ = co.bn_act_train(x2, p, Bn * H2 * W2, r2.bn, _bn_momentum(r2.bn))
                    p = part_for(Bn * H2 * W2, r3.conv.out_channels)
                    x3 = co.conv_fwd(y2, r3.wt(tok), 1, 0, p)
                    if rd is not None:
                        pd = part_for(Bn * H2 * W2, rd.conv.out_channels)
                        xd = co.conv_fwd(xin, rd.wt(tok), rd.stride, 0, pd)
                        yout, ybits, c3, cd = co.bn_act_train(x3, p, Bn * H2 * W2, r3.bn, _bn_momentum(r3.bn), res=xd,
                                                             res_stats=(pd, rd.bn, _bn_momentum(rd.bn)), want_bits=True)
                    else:
                        xd = cd = None
                        yout, ybits, c3, _ = co.bn_act_train(x3, p, Bn * H2 * W2, r3.bn, _bn_momentum(r3.bn), res=xin, want_bits=True)
                    blocks.append((xin, x1, c1, y1, x2, c2, y2, x3, c3, xd, cd, yout, xin_bits))
                    x, xin_bits = yout, ybits
                    continue
                if use_fold:
                    y1, _, c1 = co.bn_act_fold(x1, p, Bn * Hin * Win, r1.bn, _bn_momentum(r1.bn))
                    p = part_for(Bn * H2 * W2, r2.conv.out_channels, one_row=True)
                    x2 = co.conv_fwd(y1, r2.wt(tok), r2.stride, 1, p)
                    y2, _, c2 = co.bn_act_fold(x2, p, Bn * H2 * W2, r2.bn, _bn_momentum(r2.bn))
                    if rd is None:
                        p = part_for(Bn * H2 * W2, r3.conv.out_channels, one_row=True)
                        x3 = co.conv_fwd(y2, r3.wt(tok), 1, 0, p)
                        xd = cd = None
                        yout, ybits, c3 = co.bn_act_fold(x3, p, Bn * H2 * W2, r3.bn, _bn_momentum(r3.bn), res=xin, want_bits=True)
                    else:                      # projection shortcut: its BatchNorm's coefficients are needed too -> the two-launch form
                        p = part_for(Bn * H2 * W2, r3.conv.out_channels)
                        x3 = co.conv_fwd(y2, r3.wt(tok), 1, 0, p)
                        c3 = _bn_coef(r3, p, Bn * H2 * W2)
                        p = part_for(Bn * H2 * W2, rd.conv.out_channels)
                        xd = co.conv_fwd(xin, rd.wt(tok), rd.stride, 0, p)
                        cd = _bn_coef(rd, p, Bn * H2 * W2)
                        yout, ybits = co.bn_act(x3, c3, res=xd, coef_res=cd, want_bits=True)
                    blocks.append((xin, x1, c1, y1, x2, c2, y2, x3, c3, xd, cd, yout, xin_bits))
                    x, xin_bits = yout, ybits
                    continue
                c1 = _bn_coef(r1, p, Bn * Hin * Win)
                y1 = co.bn_act(x1, c1)
                p = part_for(Bn * H2 * W2, r2.conv.out_channels)
                x2 = co.conv_fwd(y1, r2.wt(tok), r2.stride, 1, p)
                c2 = _bn_coef(r2, p, Bn * H2 * W2)
                y2 = co.bn_act(x2, c2)
                p = part_for(Bn * H2 * W2, r3.conv.out_channels)
                x3 = co.conv_fwd(y2, r3.wt(tok), 1, 0, p)
                c3 = _bn_coef(r3, p, Bn * H2 * W2)
                if rd is not None:
                    p = part_for(Bn * H2 * W2, rd.conv.out_channels)
                    xd = co.conv_fwd(xin, rd.wt(tok), rd.stride, 0, p)
                    cd = _bn_coef(rd, p, Bn * H2 * W2)
                    yo = co.bn_act(x3, c3, res=xd, coef_res=cd, want_bits=want_bits)
                else:
                    xd = cd = None
                    yo = co.bn_act(x3, c3, res=xin, want_bits=want_bits)
                yout, ybits = yo if want_bits else (yo, None)
                blocks.append((xin, x1, c1, y1, x2, c2, y2, x3, c3, xd, cd, yout, xin_bits))
                x, xin_bits = yout, ybits
        if getattr(enc, "lazy_output", False):
            # models.py:39-41's dense f32 tensor is ALLOCATED here and written on first access (LazyEncoderOut): ppv_amd's own consumers
            # read the 8 x 8 map instead
            E_ = enc.enc_image_size
            out = torch.empty((x.shape[0], E_, E_, x.shape[3]), dtype=torch.float32, device=dev)
            ready = torch.cuda.Event()
            ready.record()
            xs = x

            def fill(out=out, xs=xs, ready=ready, E_=E_):
                cur = torch.cuda.current_stream(out.device)
                cur.wait_event(ready)
                with torch.cuda.device(out.device):
                    co.adaptive_pool_fwd(xs, E_, out=out)
                out.record_stream(cur); xs.record_stream(cur)
            object.__setattr__(enc, "_last_fill", fill)   # picked up by Encoder.forward right after apply() (works under no_grad too)
        else:
            out = co.adaptive_pool_fwd(x, enc.enc_image_size)
        if train:
            torch._foreach_add_(enc._nbt, 1)
        ctx.enc, ctx.saved, ctx.blocks, ctx.train, ctx.tok = enc, saved, blocks, train, tok
        ctx.holder, ctx.cells = holder, (x if holder is not None else None)
        if holder is not None and getattr(enc, "_debug_block_grads", None) is not None:
            stem_ = []                                                                              # tests: what the per-kernel path would have kept
            ctx.blocks = [t for t, _ in _tx.block_taps(holder.lease.plan, holder.lease, x, stem_)]
            ctx.saved = {"stem": stem_[0]}
        ctx.img_shape, ctx.last_hw = images.shape, (x.shape[1], x.shape[2])
        ctx.set_materialize_grads(False)
        # second output: the map the pool up-sampled (a view of the last block's output, so that returning it does not
        # alias a saved tensor): ppv_amd.decoder works on it directly and sends its gradient here (decoder.py "compact path")
        return out, x.view(x.shape)

    @staticmethod
    def backward(ctx, g_out, g_cells=None):
        enc = ctx.enc
        dev0 = (g_out if g_out is not None else g_cells).device
        if not ctx.train:
            # eval-mode BatchNorm (running statistics are constants; outside the reference's use -- validate() runs under no_grad,
            # train.py:355-451 -- but a saliency / attack pass through the frozen trunk needs it): the same launches with an infinite
            # sample count, which removes the batch-mean terms of the train-mode formula (convops.eval_bn: thread-local)
            with co.eval_bn():
                return _TrunkFn._backward(ctx, g_out, g_cells)
        return _TrunkFn._backward(ctx, g_out, g_cells)

    @staticmethod
    def _backward(ctx, g_out, g_cells=None):
        enc = ctx.enc
        dev0 = (g_out if g_out is not None else g_cells).device
        grads = {}
        tok = ctx.tok
        if ctx.holder is not None:
            # the plan executor's backward (csrc/trunk_plan.hip): one FFI call (one per gradient bucket in data-parallel runs)
            from . import trunk_exec as _tx
            import os as _os
            side = None
            if _os.environ.get("PPV_WGRAD_SIDE", "1") != "0":
                side = getattr(enc, "_wgrad_stream", None)
                if side is None:
                    side = _make_side_stream(dev0)
                    object.__setattr__(enc, "_wgrad_stream", side)
            g_img, gd = _tx.backward(enc, ctx.holder, ctx.cells, g_out, g_cells, ctx.needs_input_grad[1], tuple(ctx.img_shape), side,
                                     _bwd_main_stream(enc, dev0))
            if enc.grad_sync is not None:
                enc.grad_sync.end_of_backward()
            if gd is None:
                return (None, g_img) + (None,) * len(enc._param_list())
            return (None, g_img) + tuple(gd.get(p) for p in enc._param_list())
        # one scratch for the per-slice wgrad slabs, sized for the largest conv of this step and reused (stream order)
        # keyed on WHICH convolutions are trainable (not how many: another fine_tune split of equal count has other shapes -- r3 advisor)
        nkey = (tuple(ctx.img_shape), tuple(i_ for i_, rec_ in enumerate(r_ for blk_ in enc._blocks for r_ in blk_)
                                            if rec_ is not None and rec_.conv.weight.requires_grad))
        ncache = enc.__dict__.setdefault("_wneed_cache", {})
        need = ncache.get(nkey)
        if need is None:
            need = 0
            for blk_, sv_ in zip(enc._blocks, ctx.blocks):
                for rec_, gy_ in zip(blk_, (sv_[1], sv_[4], sv_[7], sv_[9])):
                    if rec_ is not None and rec_.conv.weight.requires_grad:
                        w_ = rec_.conv.weight
                        need = max(need, co.wgrad_scratch_bytes(gy_.numel() // gy_.shape[-1], w_.shape[0], rec_.k, rec_.k, w_.shape[1]))
            ncache[nkey] = need
        wstride = (max(need, 16) + 255) & ~255
        # three regions: ppv_bottleneck_bwd keeps the slabs of a block's three weight gradients apart and reduces them with one launch
        wscratch = torch.empty(3 * wstride, dtype=torch.uint8, device=dev0)

        # one zeroed pool for every BN's [32][2][C] backward partial sums of this step
        bn_ch = sum(r.conv.out_channels for blk_ in enc._blocks for r in blk_ if r is not None) + 64
        bpool = torch.zeros(64 * bn_ch, dtype=torch.float32, device=dev0)
        boff = [0]

        # Weight gradients run on a side stream beside the data-gradient / BN chain they do not feed: the wgrad kernels hold one
        # workgroup per CU at 20-30 % MFMA occupancy, the BN kernels next to them are pure streaming (+2.6 % whole step).
        # PPV_WGRAD_SIDE=0 keeps everything on one stream.
        import os as _os
        side = None
        if _os.environ.get("PPV_WGRAD_SIDE", "1") != "0":
            side = getattr(enc, "_wgrad_stream", None)
            if side is None:
                side = _make_side_stream(dev0)
                object.__setattr__(enc, "_wgrad_stream", side)
            side.wait_stream(torch.cuda.current_stream())
        main_stream = torch.cuda.current_stream()

        def bn_part(C):
            v = bpool[boff[0]:boff[0] + 64 * C]
            boff[0] += 64 * C
            return v

        # Data-parallel runs (enc.grad_sync): the trainable gradients live in pre-flattened buckets (dist_sync.GradSync.attach);
        # the kernels below write straight into a parameter's slice, the trunk sets param.grad to that slice itself and returns
        # nothing for it to autograd, and a bucket is all-reduced (side stream) as soon as its last slice is written.  With
        # gradients already accumulated in .grad (zero_grad(set_to_none=False), micro-batching) the slices cannot become the
        # gradient: plain tensors are produced, averaged on the current stream at the end, and returned to autograd.
        sync = enc.grad_sync
        bucketed = False
        if sync is not None:
            order = []
            for blk_ in reversed(enc._blocks):
                for rec_ in (blk_[2], blk_[1], blk_[0], blk_[3]):
                    if rec_ is not None:
                        order += [rec_.bn.weight, rec_.bn.bias, rec_.conv.weight]
            order = [p_ for p_ in order if p_.requires_grad]
            bucketed = all(p_.grad is None for p_ in order)
            if bucketed:
                sync.attach(order)
        loose = []                                            # gradients to average on the current stream (non-bucketed sync)

        def deliver(p, t):
            if bucketed:
                p.grad = t                                    # t IS the bucket slice
                sync.mark_ready(p)
            else:
                grads[p] = t
                if sync is not None:
                    loose.append(t)

        # Same-shape 1x1 / unit-stride weight gradients (conv1 / conv3 of a layer's bottlenecks: 22 + 23 in layer 3) are collected and
        # computed by ONE launch per shape without split-M slabs (co.conv_wgrad_group) once the layer's data-gradient chain has
        # produced them all; PPV_WGRAD_GROUP=0 computes each where it arises (the round-1 form), =n sets the smallest group.
        group_min = int(_os.environ.get("PPV_WGRAD_GROUP", "0"))
        pending = {}

        def _gkey(rec, gx):
            return (rec.conv.in_channels, rec.conv.out_channels, gx.shape[1] * gx.shape[2])

        w_counts = {}
        if group_min:
            for blk_, sv_ in zip(enc._blocks, ctx.blocks):
                for rec_, gy_ in zip(blk_, (sv_[1], sv_[4], sv_[7], sv_[9])):
                    if rec_ is not None and rec_.conv.weight.requires_grad and rec_.k == 1 and rec_.stride == 1 \
                            and rec_.conv.in_channels % 128 == 0 and rec_.conv.out_channels % 128 == 0:
                        k_ = _gkey(rec_, gy_)
                        w_counts[k_] = w_counts.get(k_, 0) + 1

        def flush_group(key):
            items = pending.pop(key, [])
            if not items:
                return
            gs, xs, ws = [i[0] for i in items], [i[1] for i in items], [i[2] for i in items]
            dsts = [sync.grad_view(w) for w in ws] if bucketed else None
            if side is not None:
                ev = torch.cuda.Event(); ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    dws = co.conv_wgrad_group(gs, xs, dsts)
                    for t in gs + xs:
                        t.record_stream(side)
                    for w, dw in zip(ws, dws):
                        deliver(w, dw)
            else:
                for w, dw in zip(ws, co.conv_wgrad_group(gs, xs, dsts)):
                    deliver(w, dw)

        # Where a block's three weight gradients are ENQUEUED on the side stream (their event is recorded at that point of the main
        # stream): point 0 = after bn3's backward apply, 1 = after conv3's data gradient, 2 = after bn2's apply, 3 = after conv2's data
        # gradient, 4 = after bn1's apply, 5 = after conv1's data gradient.  PPV_WGRAD_SCHED = three digits for (conv3, conv2, conv1);
        # "024" launches each as soon as its operand exists (rounds 1-2).
        wsched = [int(ch, 16) for ch in _os.environ.get("PPV_WGRAD_SCHED", "024")]   # hex digits: 6..b = the next block's points
        wsched = [max(wsched[0], 0), max(wsched[1], 2), max(wsched[2], 4)]
        cur_point = [0]
        deferred = []

        def at_point(k):
            cur_point[0] = k
            if deferred:
                keep = []
                for pt, fn in deferred:
                    if pt <= k:
                        fn()
                    else:
                        keep.append((pt, fn))
                deferred[:] = keep

        def conv_bn_bwd(rec, gy, y, xraw, coef, xin, relu, want_gpre=False, sums=None, sums2=None, wslot=None):
            trainable = rec.conv.weight.requires_grad
            affine = rec.bn.weight.requires_grad
            oa = (sync.grad_view(rec.bn.weight).view(-1), sync.grad_view(rec.bn.bias).view(-1)) if (bucketed and affine) else None
            gx, gpre, dg, db = co.bn_bwd(gy, y, xraw, coef, relu, want_gpre=want_gpre, want_affine=affine,
                                         part=bn_part(xraw.shape[-1]) if sums is None else sums, part_ready=sums is not None,
                                         sums2=sums2, out_affine=oa)
            if trainable and group_min and rec.k == 1 and rec.stride == 1 and w_counts.get(_gkey(rec, gx), 0) >= group_min:
                # one of a layer's many same-shape 1x1 convolutions: its weight gradient waits for the others (flush_groups)
                key = _gkey(rec, gx)
                for k_ in [k_ for k_ in pending if k_[2] != key[2]]:
                    flush_group(k_)                            # a key of another resolution: that layer is done
                pending.setdefault(key, []).append((gx, xin, rec.conv.weight))
                if len(pending[key]) == 24:
                    flush_group(key)
            elif trainable:
                w = rec.conv.weight
                dst = sync.grad_view(w) if bucketed else None

                def launch_w(rec=rec, gx=gx, xin=xin, w=w, dst=dst):
                    if side is not None and bucketed and co.PROFILE is None:
                        # data-parallel fast path: the kernel writes the bucket slice on the side stream (launched there by pointer), the
                        # bucket's all-reduce waits for an event recorded on that stream
                        ev = torch.cuda.Event(); ev.record(main_stream)
                        side.wait_event(ev)
                        co.conv_wgrad(gx, xin, rec.k, rec.k, rec.stride, rec.pad, scratch=wscratch, out=dst, stream=side)
                        gx.record_stream(side); xin.record_stream(side)
                        w.grad = dst
                        sync.mark_ready(w, stream=side)
                    elif side is not None and sync is None and co.PROFILE is None:
                        # single-process fast path: the launch goes to the side stream by pointer (the stream context manager and the
                        # current-stream look-ups cost the host ~25 us per weight gradient, 93 per step)
                        ev = torch.cuda.Event(); ev.record(main_stream)
                        side.wait_event(ev)
                        dw = torch.empty(w.shape, dtype=torch.float32, device=w.device)     # caching allocator: owned by the CURRENT stream ...
                        co.conv_wgrad(gx, xin, rec.k, rec.k, rec.stride, rec.pad, scratch=wscratch, out=dw, stream=side)
                        gx.record_stream(side); xin.record_stream(side); dw.record_stream(side)   # ... so its reuse waits for the side stream
                        deliver(w, dw)
                    elif side is not None:
                        ev = torch.cuda.Event(); ev.record()
                        with torch.cuda.stream(side):
                            side.wait_event(ev)
                            dw = co.conv_wgrad(gx, xin, rec.k, rec.k, rec.stride, rec.pad, scratch=wscratch, out=dst)
                            gx.record_stream(side); xin.record_stream(side)
                            deliver(w, dw)
                    else:
                        deliver(w, co.conv_wgrad(gx, xin, rec.k, rec.k, rec.stride, rec.pad, scratch=wscratch, out=dst))
                slot = {3: 0, 2: 1, 1: 2}.get(wslot, None)
                if slot is None or wsched[slot] <= cur_point[0]:
                    launch_w()
                else:
                    deferred.append((wsched[slot], launch_w))
            if affine:
                deliver(rec.bn.weight, dg)
                deliver(rec.bn.bias, db)
            return gx, gpre

        # Gradients between blocks travel PRE-MASKED by the ReLU of the tensor they belong to: the mask of a block's output
        # (yout > 0) is applied where that gradient is produced (adaptive-pool backward for the last block, the conv1
        # data-gradient store of the following block otherwise).  bn3 / downsample-bn backward then need neither yout
        # nor a separate masked copy: -2 tensors of the 4C-wide size per block against one extra mask read.
        last = ctx.blocks[-1][11]
        g = co.adaptive_pool_bwd(g_out.contiguous(), ctx.last_hw, relu_of=last) if g_out is not None else None
        if g_cells is not None:                              # gradient that arrived on the un-pooled map: mask + bf16 (E == H: 1x1 windows)
            gc = co.adaptive_pool_bwd(g_cells.contiguous(), ctx.last_hw, relu_of=last)
            g = gc if g is None else g.add_(gc)
        taps = getattr(enc, "_debug_block_grads", None)      # tests: per-block (g_out, g_in) taps, last block first
        # The conv1 data-gradient launch that stores a block's input gradient also takes the sums bn3-backward of the PREVIOUS
        # block needs from it (sum g, sum g * x3 per channel): that BN's reduce pass over g and x3 (2 tensors of the 4C-wide
        # size) becomes one read of x3 in the store loop.  PPV_DGRAD_BNRED=0 keeps the separate passes, 1 fuses bn3 only, 2 adds bn1 / bn2,
        # 3 (default) also the projection shortcuts' sums (taken by bn3's apply pass).
        red_level = int(_os.environ.get("PPV_DGRAD_BNRED", "3"))
        fuse_red, fuse_red12, fuse_proj = red_level >= 1, red_level >= 2, red_level >= 3
        order = list(zip(reversed(enc._blocks), reversed(ctx.blocks)))
        sums3 = None
        # PPV_BLOCK_EXEC (default on): the twelve launches + three stream forks of a bottleneck without projection go through ONE FFI
        # crossing (csrc/block_exec.hip ppv_bottleneck_bwd: the calls below, same order, same arguments).  Only in the default
        # configuration: single process, no taps / per-class timing, default weight-gradient schedule.
        fast_bwd = (_os.environ.get("PPV_BLOCK_EXEC", "1") != "0" and (sync is None or (bucketed and side is not None)) and taps is None
                    and co.PROFILE is None and not group_min and wsched == [0, 2, 4] and red_level == 3
                    and ctx.train)       # (eval-mode BatchNorm backward goes through convops.eval_bn(): the per-kernel calls)
        if fast_bwd:
            bwa = _lib.BottleneckBwd()
            bwa.zero_page = co.zero_page(dev0).data_ptr()
            bwa.wscratch = wscratch.data_ptr()
            # PPV_WGRAD_REDUCE3=1 (opt-in): one slab reduce per block instead of three (ppv_conv_wgrad_ex / ppv_wgrad_reduce_multi: 58
            # launches fewer on the side stream).  MEASURED: 5680 / 5680 with against 5686 / 5693 images/s without -- the reduces were
            # never on the critical path, and the slabs of the first two gradients now stay live until the third is done.
            bwa.wstride = wstride if _os.environ.get("PPV_WGRAD_REDUCE3", "0") == "1" else 0
            kc_all = torch.empty(3 * 3 * 2048 + 64, dtype=torch.float32, device=dev0)     # coefficient scratch of the three BatchNorms (stream-ordered reuse)
            side_ptr = side.cuda_stream if side is not None else None
        for bi, (blk, sv) in enumerate(order):
            g_blk_out = g
            r1, r2, r3, rd = blk
            if fast_bwd and rd is None and r2.stride == 1 and 4 * r1.conv.out_channels <= 2048:
                P_ = r1.conv.out_channels
                xin = sv[0]
                Bn, Hh, Ww, _ = xin.shape
                Mr = Bn * Hh * Ww
                a = bwa
                if isinstance(sv, _BlockSave):
                    ap_, ao_ = sv.arena.data_ptr(), sv.off
                    a.x1, a.y1, a.x2, a.y2, a.x3 = ap_ + ao_[0], ap_ + ao_[1], ap_ + ao_[2], ap_ + ao_[3], ap_ + ao_[4]
                    a.c1, a.c2, a.c3 = ap_ + ao_[6], ap_ + ao_[7], ap_ + ao_[8]
                    a.xin_bits = sv.prev_ptr
                    keep_side = (sv.arena, xin)                               # what the side stream reads of the saved tensors
                else:
                    _, x1, c1, y1, x2, c2, y2, x3, c3, _, _, _, xin_bits = sv
                    a.x1, a.y1, a.x2, a.y2, a.x3 = x1.data_ptr(), y1.data_ptr(), x2.data_ptr(), y2.data_ptr(), x3.data_ptr()
                    a.c1, a.c2, a.c3 = c1.data_ptr(), c2.data_ptr(), c3.data_ptr()
                    a.xin_bits = xin_bits.data_ptr() if xin_bits is not None else None
                    keep_side = (y2, y1, xin)
                a.g, a.xin = g.data_ptr(), xin.data_ptr()
                a.wd1, a.wd2, a.wd3 = r1.wd(tok).data_ptr(), r2.wd(tok).data_ptr(), r3.wd(tok).data_ptr()
                if sums3 is not None:
                    a.part3, a.part3_ready = sums3.data_ptr(), 1
                else:
                    a.part3, a.part3_ready = bn_part(4 * P_).data_ptr(), 0
                a.red2 = a.red1 = int(co.red_supported(Mr, P_))
                pb = bpool.data_ptr() + 4 * boff[0]
                boff[0] += 128 * P_
                a.part2, a.part1 = pb, pb + 256 * P_
                kb = kc_all.data_ptr()
                a.kc3, a.kc2, a.kc1 = kb, kb + 4 * 3 * 2048, kb + 8 * 3 * 2048
                # the five gradients that never leave the block in ONE buffer (gx3 | gy2 | gx2 | gy1 | gx1), the block-input gradient a tensor
                tb, Tb = (Mr * P_ * 2 + 255) & ~255, Mr * 8 * P_
                work = torch.empty(Tb + 4 * tb, dtype=torch.uint8, device=dev0)
                wp = work.data_ptr()
                gin = torch.empty_like(xin)
                a.gx3, a.gy2, a.gx2, a.gy1, a.gx1, a.gin = wp, wp + Tb, wp + Tb + tb, wp + Tb + 2 * tb, wp + Tb + 3 * tb, gin.data_ptr()
                outs, wouts = [], []
                for i_, r_, C_ in ((3, r3, 4 * P_), (2, r2, P_), (1, r1, P_)):
                    if r_.bn.weight.requires_grad:
                        if bucketed:                     # the kernels write the parameters' slices of the flat gradient buckets
                            dg_, db_ = sync.grad_view(r_.bn.weight).view(-1), sync.grad_view(r_.bn.bias).view(-1)
                        else:
                            dg_ = torch.empty(C_, dtype=torch.float32, device=dev0)
                            db_ = torch.empty(C_, dtype=torch.float32, device=dev0)
                        setattr(a, "dg%d" % i_, dg_.data_ptr()); setattr(a, "db%d" % i_, db_.data_ptr())
                        outs += [(r_.bn.weight, dg_), (r_.bn.bias, db_)]
                    else:
                        setattr(a, "dg%d" % i_, None); setattr(a, "db%d" % i_, None)
                    w_ = r_.conv.weight
                    if w_.requires_grad:
                        if bucketed:
                            dw_ = sync.grad_view(w_)
                            wouts.append((w_, dw_))
                        else:
                            dw_ = torch.empty(w_.shape, dtype=torch.float32, device=dev0)
                            outs.append((w_, dw_))
                            if side is not None:
                                dw_.record_stream(side)
                        setattr(a, "dw%d" % i_, dw_.data_ptr())
                    else:
                        setattr(a, "dw%d" % i_, None)
                sums3 = None
                a.x3_prev = a.part3_prev = None
                if bi + 1 < len(order):
                    svn = order[bi + 1][1]
                    if isinstance(svn, _BlockSave):
                        Cn, rows_n, x3p = 4 * svn.shp[3], svn.shp[0] * svn.shp[1] * svn.shp[2], svn.ptr(4)
                    else:
                        x3n = svn[7]
                        Cn, rows_n, x3p = x3n.shape[-1], x3n.numel() // x3n.shape[-1], x3n.data_ptr()
                    if co.red_supported(rows_n, Cn):
                        sums3 = bn_part(Cn)
                        a.x3_prev, a.part3_prev = x3p, sums3.data_ptr()
                a.B, a.H, a.W, a.planes = Bn, Hh, Ww, P_
                _lib.check(_lib.lib().ppv_bottleneck_bwd(_lib.ctypes.byref(a), main_stream.cuda_stream, side_ptr), "ppv_bottleneck_bwd")
                if side is not None:                   # operands the side stream reads: the allocator must not recycle them before it has
                    work.record_stream(side)
                    for t_ in keep_side:
                        t_.record_stream(side)
                for p_, t_ in outs:
                    deliver(p_, t_)
                for w_, dw_ in wouts:                  # bucket slices written on the side stream: the bucket's all-reduce waits for it
                    w_.grad = dw_
                    sync.mark_ready(w_, stream=side)
                g = gin
                continue
            xin, x1, c1, y1, x2, c2, y2, x3, c3, xd, cd, yout, xin_bits = sv
            hw_in, hw_mid = (xin.shape[1], xin.shape[2]), (y2.shape[1], y2.shape[2])
            # a down-sampling block's projection BatchNorm sees the same gradient as bn3: bn3's apply pass takes its sums too
            sumsd = bn_part(xd.shape[-1]) if (fuse_proj and rd is not None) else None
            deferred[:] = [(pt - 6, fn) for pt, fn in deferred]    # points 6..11 of a block = points 0..5 of the next one
            at_point(0)
            gx3, _ = conv_bn_bwd(r3, g, None, x3, c3, y2, 0, sums=sums3, sums2=None if sumsd is None else (xd, sumsd), wslot=3)
            red = sums3 = None
            if fuse_red and bi + 1 < len(order):
                x3_prev = order[bi + 1][1][7]                # raw conv3 output of the block this gradient flows into
                if co.red_supported(x3_prev.numel() // x3_prev.shape[-1], x3_prev.shape[-1]):
                    sums3 = bn_part(x3_prev.shape[-1])
                    red = (x3_prev, sums3)
            # bn2 / bn1 (BN + ReLU, no residual): the data-gradient launch recomputes the ReLU mask from the raw conv output,
            # stores the masked gradient and takes the BN-backward sums (64-column tensors only at >= 128 Ki pixels: the 128 x 64 tile)
            if fuse_red12 and co.red_supported(x2.numel() // x2.shape[-1], x2.shape[-1]):
                s2 = bn_part(x2.shape[-1])
                gy2 = co.conv_dgrad(gx3, r3.wd(tok), 1, 0, hw_mid, red=(x2, s2, c2))
                at_point(1); at_point(2)
                gx2, _ = conv_bn_bwd(r2, gy2, None, x2, c2, y1, 0, sums=s2, wslot=2)
            else:
                gy2 = co.conv_dgrad(gx3, r3.wd(tok), 1, 0, hw_mid)
                at_point(1); at_point(2)
                gx2, _ = conv_bn_bwd(r2, gy2, None, x2, c2, y1, 2, wslot=2)   # mask recomputed from x2 (no residual): y2 not read
            if fuse_red12 and co.red_supported(x1.numel() // x1.shape[-1], x1.shape[-1]):
                s1 = bn_part(x1.shape[-1])
                gy1 = co.conv_dgrad(gx2, r2.wd(tok), r2.stride, 1, hw_in, red=(x1, s1, c1))
                at_point(3); at_point(4)
                gx1, _ = conv_bn_bwd(r1, gy1, None, x1, c1, xin, 0, sums=s1, wslot=1)
            else:
                gy1 = co.conv_dgrad(gx2, r2.wd(tok), r2.stride, 1, hw_in)
                at_point(3); at_point(4)
                gx1, _ = conv_bn_bwd(r1, gy1, None, x1, c1, xin, 2, wslot=1)
            if rd is not None:
                gxd, _ = conv_bn_bwd(rd, g, None, xd, cd, xin, 0, sums=sumsd)
                gin = co.conv_dgrad(gxd, rd.wd(tok), rd.stride, 0, hw_in)
                g = co.conv_dgrad(gx1, r1.wd(tok), 1, 0, hw_in, addend=gin, relu_bits=xin_bits, red=red)
            else:
                g = co.conv_dgrad(gx1, r1.wd(tok), 1, 0, hw_in, addend=g, relu_bits=xin_bits, red=red)
            at_point(5)
            if taps is not None:
                taps.append((g_blk_out, g))
        g_img = None
        needs_img = ctx.needs_input_grad[1]
        for k_ in list(pending):
            flush_group(k_)
        at_point(1 << 30)                                     # whatever is still deferred
        if needs_img or enc._stem.bn.weight.requires_grad or enc._stem.conv.weight.requires_grad:
            raw0, c0, y0, arg0 = ctx.saved["stem"]
            st = enc._stem
            if st.conv.weight.requires_grad:
                raise NotImplementedError("the stem convolution is frozen in the reference (models.py:43-54)")
            if _os.environ.get("PPV_STEM_BWD_FUSED", "1") != "0" and raw0.shape[-1] <= 256:
                # pooled gradient -> gradient of the raw stem output in two passes over the pooled tensors (no 268-MB pre-pool tensor)
                gx0, dg, db = co.maxpool_bn_bwd(g, y0, arg0, raw0, c0, want_affine=st.bn.weight.requires_grad)
            else:
                gpre0 = co.maxpool_relu_bwd(g, y0, arg0, (raw0.shape[1], raw0.shape[2]))
                gx0, _, dg, db = co.bn_bwd(gpre0, None, raw0, c0, False, want_affine=st.bn.weight.requires_grad, part=bn_part(64))
            if st.bn.weight.requires_grad:
                grads[st.bn.weight], grads[st.bn.bias] = dg, db
                if sync is not None:
                    loose += [dg, db]
            if needs_img:
                g_img = co.stem_dgrad(gx0, st.wd(tok))
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        if sync is not None:
            if loose:
                sync.reduce_now(loose)           # accumulation mode: averaged in stream order, then handed to autograd
            sync.end_of_backward()               # tail bucket out; joins the all-reduce stream unless the harness defers it (flush())
        return (None, g_img) + tuple(grads.get(p) for p in enc._param_list())


class Encoder(nn.Module):
    """Encoder (models.py:8-54).  ``layers`` (extra, keyword) shrinks the trunk for tests; default = ResNet-101."""

    def __init__(self, encoded_image_size=36, *, layers=LAYERS, lazy_output=True, precision="bf16"):
        super().__init__()
        if precision not in ("bf16", "fp32"):
            raise ValueError("ppv_amd Encoder: precision is 'bf16' (bf16 storage, f32 accumulate: BASELINE configs 3 / 5) or 'fp32' (the "
                             "reference's own precision, models.py:31-41 under train.py:245)")
        self.precision = precision         # "fp32": f32 activations end to end, every convolution at f32 level on the MFMA kernels (six bf16
        #                                    products of a three-way operand split), BatchNorm / ReLU / residual / pools on csrc/bn_f32.hip
        self.enc_image_size = encoded_image_size
        self.lazy_output = lazy_output     # the dense [B,E,E,2048] f32 output is written on first access (LazyEncoderOut)
        # the reference loads ImageNet weights (models.py:17); offline there are none: torchvision's random init
        self.resnet = _make_trunk(layers)
        self.adaptive_pool = nn.AdaptiveAvgPool2d((encoded_image_size, encoded_image_size))
        self._index()
        self._step_token = 0
        self.grad_sync = None      # ppv_amd.dist_sync.GradSync for data-parallel training (bench.py / DDP harness)
        self.fine_tune()

    # whole-module pickling (the reference resumes with `encoder = checkpoint['encoder']`, train.py:145): streams, cached bf16
    # weight layouts and the per-conv records are runtime state, rebuilt on load
    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ("_stem", "_blocks", "_wl", "_wl_prefetched", "_wgrad_stream", "grad_sync", "_debug_block_grads", "_plist_cache", "_plist_mid", "_last_fill", "_grad_wanted", "_nbt_cache", "_wneed_cache", "_plans", "_bwd_main_stream"):
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self.grad_sync = None
        self._index()

    def _index(self):
        r = self.resnet
        object.__setattr__(self, "_stem", _ConvRec(r[0], r[1], stem=True))
        blocks = []
        for li in range(4, 8):
            for b in r[li]:
                rd = _ConvRec(b.downsample[0], b.downsample[1]) if b.downsample is not None else None
                blocks.append((_ConvRec(b.conv1, b.bn1), _ConvRec(b.conv2, b.bn2), _ConvRec(b.conv3, b.bn3), rd))
        object.__setattr__(self, "_blocks", blocks)

    def _stat_pool_elems(self, B, H, W):
        """f32 elements of the per-step BN partial-sum pool for a [B,3,H,W] input."""
        tot = co.stat_tiles(B * (H // 2) * (W // 2)) * 2 * 64
        h, w = H // 4, W // 4
        for r1, r2, r3, rd in self._blocks:
            tot += co.stat_tiles(B * h * w) * 2 * r1.conv.out_channels
            h, w = h // r2.stride, w // r2.stride
            for r in (r2, r3, rd):
                if r is not None:
                    tot += co.stat_tiles(B * h * w) * 2 * r.conv.out_channels
        return tot

    def _refresh_weight_layouts(self):
        """(Re)build the shared bf16 layouts of the trainable convs and refresh them with one launch."""
        recs = [r for blk in self._blocks for r in blk if r is not None and r.conv.weight.requires_grad and r.conv.weight.is_cuda]
        wl = getattr(self, "_wl", None)
        if not recs:
            for blk in self._blocks:
                for r in blk:
                    if r is not None:
                        r.wl = None
            object.__setattr__(self, "_wl", None)
            return
        if wl is None or len(wl.weights) != len(recs) or not wl.valid() or any(a is not r.conv.weight for a, r in zip(wl.weights, recs)):
            for blk in self._blocks:
                for r in blk:
                    if r is not None:
                        r.wl = None
            wl = co.WeightLayouts([r.conv.weight for r in recs])
            for i, r in enumerate(recs):
                r.wl, r.wl_idx = wl, i
            object.__setattr__(self, "_wl", wl)
        wl.refresh()

    def invalidate_weight_cache(self):
        """Drop the cached bf16 layouts of the FROZEN convolutions (stem, layer1, everything under fine_tune(False)).  They are
        keyed on the parameter's version counter and storage pointer, which in-place writes through ``p.data`` (EMA, weight
        surgery) do not change; train() / eval() / load_state_dict() call this, call it yourself after such a write."""
        self.__dict__["_wcache_gen"] = self.__dict__.get("_wcache_gen", 0) + 1      # the plan executor's pointer tables follow
        self._stem.invalidate()
        for blk in self._blocks:
            for r in blk:
                if r is not None:
                    r.invalidate()

    def train(self, mode=True):
        self.invalidate_weight_cache()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_weight_cache()
        self.__dict__["_plist_gen"] = self.__dict__.get("_plist_gen", 0) + 1      # assign=True replaces the Parameter objects
        return out

    def prefetch_weight_layouts(self):
        """Optional: convert the trainable conv weights to their bf16 GEMM layouts NOW, on the current stream, for the next
        forward (which otherwise does it first thing).  Call it right after ``optimizer.step()`` -- e.g. on the stream the
        optimizer ran on, beside other work -- and make the stream that runs the forward wait for this one.  Weights changed
        after this call and before the next forward are not seen by that forward."""
        self._refresh_weight_layouts()
        object.__setattr__(self, "_wl_prefetched", True)

    def _param_list(self):
        # cached: walking the module tree costs ~0.8 ms and the trunk calls this twice per step; the trunk's structure is fixed after
        # __init__ (Parameter OBJECTS are replaced by load_state_dict(assign=True) / .to_empty(): the cache is keyed on their ids)
        cached = self.__dict__.get("_plist_cache")
        if cached is not None and cached[0] == self.__dict__.get("_plist_gen", 0):
            lst = cached[1]
            # cheap validation on every call: a PARENT module's load_state_dict(assign=True) or direct weight surgery replaces
            # Parameter objects without passing through this class's hooks (r2 advisor) -- sentinels from the stem, the first
            # trainable block and the last block must still be the cached objects
            r, sent = self.resnet, self.__dict__.get("_plist_mid")
            if sent is not None and sent[0] is r[0].weight and sent[1] is r[5][0].conv1.weight and sent[2] is r[7][-1].bn3.bias:
                return lst
        lst = list(self.resnet.parameters())
        self.__dict__["_plist_mid"] = (self.resnet[0].weight, self.resnet[5][0].conv1.weight, self.resnet[7][-1].bn3.bias)
        self.__dict__["_plist_cache"] = (self.__dict__.get("_plist_gen", 0), lst)
        return lst

    def _apply(self, fn, *a, **kw):                       # .cuda() / .to() / .float(): parameters may be new objects afterwards
        self.__dict__["_plist_gen"] = self.__dict__.get("_plist_gen", 0) + 1
        return super()._apply(fn, *a, **kw)

    @property
    def _nbt(self):
        # the 104 num_batches_tracked buffers; cached per _apply generation (walking the module tree costs 0.4 ms of host time per step),
        # validated by two sentinels (a buffer re-assigned from outside makes the list stale)
        c = self.__dict__.get("_nbt_cache")
        gen = self.__dict__.get("_plist_gen", 0)
        if c is not None and c[0] == gen and c[1][0] is self.resnet[1].num_batches_tracked and c[1][-1] is self.resnet[7][-1].bn3.num_batches_tracked:
            return c[1]
        lst = [m.num_batches_tracked for m in self.resnet.modules() if isinstance(m, nn.BatchNorm2d)]
        self.__dict__["_nbt_cache"] = (gen, lst)
        return lst

    def forward(self, images):
        if not images.is_cuda:
            raise RuntimeError("ppv_amd Encoder runs on an MI355X (images must be a cuda tensor); no CPU path")
        if getattr(self, "precision", "bf16") == "fp32":
            return self._forward_fp32(images)
        object.__setattr__(self, "_grad_wanted", torch.is_grad_enabled())      # (autograd.Function.forward runs with grad mode off)
        out, cells = _TrunkFn.apply(self, images, *self._param_list())
        if getattr(self, "lazy_output", False):
            fill = self.__dict__.pop("_last_fill", None)
            out = LazyEncoderOut.wrap(out, fill)
        out._ppv_cells = cells             # the 8x8 map behind the up-sampled output (consumed by ppv_amd.decoder, ignored otherwise)
        return out

    # ------------------------------------------------------------------ the fp32 product mode (round 6)
    def _forward_fp32(self, images, taps=None):
        """Encoder(precision="fp32").forward: the reference's precision (models.py:31-41; train.py:245 trains the trunk in fp32) as a product
        path.  f32 NHWC activations end to end; every convolution on the hand-written MFMA kernels at f32 level -- forward, data gradient
        and weight gradient as six bf16 products of a three-way operand split (ppv_amd.nn_ops.conv2d_f32(exact=True), ~2^-24 per product);
        BatchNorm (batch statistics in train mode WITH the running-statistics update, running statistics in eval mode) + residual + ReLU,
        the stem's max-pool and the adaptive average pool on csrc/bn_f32.hip (ppv_bn_f32_*, ppv_maxpool_f32_*, ppv_adaptive_pool_f32_*):
        no torch element-wise op between the convolutions.  Differentiable through autograd (custom Functions per layer); deterministic
        element-wise side.  ~10x the bf16 path's time per image: the precision mode, not the throughput mode.
        ``taps``: optional list that receives every block's output [B,H,W,C] f32 (stem pool output first)."""
        from .nn_ops import adaptive_avg_pool_f32, batch_norm_f32, conv2d_f32, max_pool3x3s2_f32
        x = torch.nn.functional.pad(images.float().permute(0, 2, 3, 1), (0, 5)).contiguous()       # NHWC f32, 3 -> 8 channels (vector split)

        def conv(t, rec, w=None):
            w = rec.conv.weight if w is None else w
            return conv2d_f32(t, w, None, rec.stride, rec.pad, weight_grad=rec.conv.weight.requires_grad, accurate_wgrad=True, exact=True)

        st = self._stem
        y = batch_norm_f32(conv(x, st, torch.nn.functional.pad(st.conv.weight, (0, 0, 0, 0, 0, 5))), st.bn)
        y = max_pool3x3s2_f32(y)
        if taps is not None:
            taps.append(y)
        for r1, r2, r3, rd in self._blocks:
            o = batch_norm_f32(conv(y, r1), r1.bn)
            o = batch_norm_f32(conv(o, r2), r2.bn)
            idn = y if rd is None else batch_norm_f32(conv(y, rd), rd.bn, relu=False)
            y = batch_norm_f32(conv(o, r3), r3.bn, res=idn)
            if taps is not None:
                taps.append(y)
        out = adaptive_avg_pool_f32(y, self.enc_image_size)
        out._ppv_cells = None          # (the compact hand-over to ppv_amd.decoder is a bf16 map: the fp32 mode hands over the dense tensor only)
        return out

    # ------------------------------------------------------------------ fp32-accurate forward (parity instrument, not the product path)
    @torch.no_grad()
    def forward_fp32_accurate(self, images, taps=None):
        """The same trunk with f32 activations end to end, for parity against the UN-ROUNDED fp32 reference (north_star:
        "conv activations within 1e-3 rel fp32"; the product stores bf16 activations, which a 101-layer train-mode-BN network at
        random initialisation amplifies far beyond that, see tests/test_encoder_fp32_gpu.py).

        Every convolution still runs on the hand-written MFMA kernel, as three bf16 products accumulated in f32:
        [x_hi | x_lo | x_hi] x [W_hi | W_hi | W_lo] (ppv_bn_act_split3 fuses BatchNorm + ReLU + the split, as in ppv_amd.fan);
        the product error drops from 2^-9 to about 2^-16.  Batch statistics, the residual add and the pools are f32 torch
        element-wise ops on the device (train-mode BatchNorm: biased variance of the batch, models.py:31-41 / train.py:245;
        running statistics are NOT updated).  Forward only.  ``taps``: optional list that receives every block's output
        [B,H,W,C] f32 (stem pool output first)."""
        from . import _lib
        from ._lib import check, ptr, stream_ptr
        if not images.is_cuda:
            raise RuntimeError("ppv_amd Encoder runs on an MI355X (images must be a cuda tensor); no CPU path")
        x = images.float().permute(0, 2, 3, 1).contiguous()                                  # NHWC f32
        train = self.resnet[1].training

        def w3(w):
            w = w.detach().float()
            hi = w.bfloat16().float()
            lo = (w - hi).bfloat16().float()
            w3_ = torch.cat([hi, hi, lo], dim=1)
            cin3 = (w3_.shape[1] + 63) // 64 * 64
            if cin3 != w3_.shape[1]:
                w3_ = torch.nn.functional.pad(w3_, (0, 0, 0, 0, 0, cin3 - w3_.shape[1]))
            return co.weight_layout(w3_.contiguous(), 0)

        def split3(t, coef, relu):
            B_, H_, W_, C_ = t.shape
            cp = (3 * C_ + 63) // 64 * 64
            y = torch.empty((B_, H_, W_, cp), dtype=torch.bfloat16, device=t.device)
            check(_lib.lib().ppv_bn_act_split3(ptr(t), ptr(coef), ptr(y), B_ * H_ * W_, C_, cp, int(relu), C_, stream_ptr()),
                  "ppv_bn_act_split3")
            return y

        def coef_of(bn, t):
            if train:
                mean = t.mean(dim=(0, 1, 2), dtype=torch.float64)
                var = (t.double() - mean).square().mean(dim=(0, 1, 2))
            else:
                mean, var = bn.running_mean.double(), bn.running_var.double()
            scale = bn.weight.detach().double() / torch.sqrt(var + bn.eps)
            return torch.stack([scale, bn.bias.detach().double() - mean * scale]).float().contiguous()

        def conv(t, coef, relu, rec):
            """act(bn(t)) -> conv, f32 in / f32 out"""
            return co.conv_fwd(split3(t, coef, relu), w3(rec.conv.weight), rec.stride, rec.pad, out_f32=True)

        st = self._stem
        x = torch.nn.functional.pad(x, (0, 5))                                               # 3 -> 8 channels (vector split)
        raw = co.conv_fwd(split3(x, None, False), w3(torch.nn.functional.pad(st.conv.weight.detach(), (0, 0, 0, 0, 0, 5))), 2, 3,
                          out_f32=True)
        c = coef_of(st.bn, raw)
        y = torch.relu(raw * c[0] + c[1])
        y = torch.nn.functional.max_pool2d(y.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous()
        if taps is not None:
            taps.append(y)
        for r1, r2, r3, rd in self._blocks:
            x1 = conv(y, None, False, r1)
            x2 = conv(x1, coef_of(r1.bn, x1), True, r2)
            x3 = conv(x2, coef_of(r2.bn, x2), True, r3)
            c3 = coef_of(r3.bn, x3)
            if rd is not None:
                xd = conv(y, None, False, rd)
                cd = coef_of(rd.bn, xd)
                idn = xd * cd[0] + cd[1]
            else:
                idn = y
            y = torch.relu(x3 * c3[0] + c3[1] + idn)
            if taps is not None:
                taps.append(y)
        return torch.nn.functional.adaptive_avg_pool2d(y.permute(0, 3, 1, 2), self.enc_image_size).permute(0, 2, 3, 1).contiguous()

    def forward_fp32_train(self, images, exact=False):
        """fp32-accurate TRAINING pass of the trunk (parity instrument, not the product path): like forward_fp32_accurate every
        convolution runs on the hand-written MFMA kernels as three bf16 products accumulated in f32 -- forward, data gradient AND
        weight gradient (ppv_amd.nn_ops.conv2d_f32, accurate_wgrad) -- with f32 activations end to end; train-mode BatchNorm (batch
        statistics, biased variance: models.py:31-41 under train.py:245), ReLU, the residual adds and the pools are f32 torch
        element-wise / reduction ops on the device, differentiated by autograd.  Running statistics are NOT updated.  images
        [B,3,H,W] -> [B,E,E,2048] f32 attached to the graph: BASELINE configs[0] (batch 4, fp32) has a reference-precision step on
        the GPU whose lens gradient can be held against the CPU reference tightly (tests/test_config0_gpu.py).  exact=True: every
        convolution as SIX bf16 products of a three-way operand split (~2^-24 per product, f32 level; nn_ops.conv2d_f32(exact=True)):
        the random-init train-mode trunk amplifies the 2^-16 of the three-product form to 1e-2 at the output."""
        from .nn_ops import conv2d_f32
        F_ = torch.nn.functional
        if not images.is_cuda:
            raise RuntimeError("ppv_amd Encoder runs on an MI355X (images must be a cuda tensor); no CPU path")
        x = F_.pad(images.float().permute(0, 2, 3, 1), (0, 5)).contiguous()                    # NHWC f32, 3 -> 8 channels (vector split)

        def bn(t, m, res=None, relu=True):
            mean = t.mean(dim=(0, 1, 2))
            var = (t - mean).square().mean(dim=(0, 1, 2))
            y = (t - mean) * (torch.rsqrt(var + m.eps) * m.weight) + m.bias
            if res is not None:
                y = y + res
            return torch.relu(y) if relu else y

        def conv(t, rec):
            w = rec.conv.weight
            return conv2d_f32(t, w, None, rec.stride, rec.pad, weight_grad=w.requires_grad, accurate_wgrad=True, exact=exact)

        st = self._stem
        w0 = F_.pad(st.conv.weight, (0, 0, 0, 0, 0, 5))
        y = bn(conv2d_f32(x, w0, None, 2, 3, weight_grad=st.conv.weight.requires_grad, accurate_wgrad=True, exact=exact), st.bn)
        y = F_.max_pool2d(y.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous()
        for r1, r2, r3, rd in self._blocks:
            o = bn(conv(y, r1), r1.bn)
            o = bn(conv(o, r2), r2.bn)
            idn = y if rd is None else bn(conv(y, rd), rd.bn, relu=False)
            y = bn(conv(o, r3), r3.bn, res=idn)
        return F_.adaptive_avg_pool2d(y.permute(0, 3, 1, 2), self.enc_image_size).permute(0, 2, 3, 1).contiguous()

    def clip_gradients_(self, grad_clip):
        """The reference's ``clip_gradient`` (Image_Caption/utils.py, called at train.py:311-316: ``p.grad.data.clamp_(-c, c)`` for every
        parameter) on this module's parameters.  The plan executor hands out the gradients as views of ONE flat f32 buffer
        (trunk_exec): when every gradient is such a view the clamp is ONE element-wise pass over that buffer instead of two
        multi-tensor passes (clamp_min, clamp_max) over 314 tensors; any other case falls back to those."""
        grads = [p.grad for p in self.parameters() if p.grad is not None]
        if not grads:
            return
        # (the tensors autograd hands to .grad share the flat buffer's storage but carry no ._base: they are recognised by storage)
        st = grads[0].untyped_storage()
        if all(g.dtype == torch.float32 and g.is_contiguous() and g.untyped_storage().data_ptr() == st.data_ptr() for g in grads):
            lo = min(g.storage_offset() for g in grads)
            hi = max(g.storage_offset() + g.numel() for g in grads)
            if sum(g.numel() for g in grads) * 2 > hi - lo:      # dense: the gaps are the few zero padding floats between slices
                torch.empty(0, dtype=torch.float32, device=grads[0].device).set_(st, lo, (hi - lo,)).clamp_(-grad_clip, grad_clip)
                return
        torch._foreach_clamp_min_(grads, -grad_clip)
        torch._foreach_clamp_max_(grads, grad_clip)

    def fine_tune(self, fine_tune=True):
        """models.py:43-54: freeze everything, then un-freeze children [5:] (layer2..4)."""
        for p in self.resnet.parameters():
            p.requires_grad = False
        for c in list(self.resnet.children())[5:]:
            for p in c.parameters():
                p.requires_grad = fine_tune
