"""Drop-in for reference ``Face-DeId/RAFT/core/corr.py:12 CorrBlock`` on MI355X: all-pairs correlation volume (fp32
MFMA), 4-level average-pool pyramid and the 9x9 bilinear window lookup, through libppv_hip.so.  Same constructor and
``__call__(coords)`` contract.  Autograd: gradients flow to ``fmap1`` / ``fmap2`` through the lookup, the pyramid and the
volume (RAFT detaches ``coords`` every iteration, raft.py:123, and the reference's own CUDA extension leaves
``coords_grad`` zero), which is what ``loss_RAFT`` (core/utils.py:437-462) back-propagates."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


class _VolumeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, num_levels):
        L = _lib.lib()
        f1 = f1.float().contiguous()
        f2 = f2.float().contiguous()
        B, C, H, W = f1.shape
        corr = torch.empty((B * H * W, 1, H, W), dtype=torch.float32, device=f1.device)
        check(L.ppv_corr_volume(ptr(f1), ptr(f2), ptr(corr), B, C, H * W, stream_ptr()), "ppv_corr_volume")
        pyr = [corr]
        h, w = H, W
        for _ in range(num_levels - 1):
            nxt = torch.empty((B * H * W, 1, h // 2, w // 2), dtype=torch.float32, device=f1.device)
            check(L.ppv_avgpool2(ptr(pyr[-1]), ptr(nxt), B * H * W, h, w, stream_ptr()), "ppv_avgpool2")
            pyr.append(nxt)
            h, w = h // 2, w // 2
        ctx.save_for_backward(f1, f2)
        ctx.set_materialize_grads(False)
        return tuple(pyr)

    @staticmethod
    def backward(ctx, *gs):
        f1, f2 = ctx.saved_tensors
        L = _lib.lib()
        B, C, H, W = f1.shape
        n = B * H * W
        # fold the pyramid gradients down to level 0 (adjoint of the avg-pool chain)
        g = [x.contiguous().clone() if x is not None else None for x in gs]
        for lvl in range(len(g) - 1, 0, -1):
            if g[lvl] is None:
                continue
            h, w = H >> (lvl - 1), W >> (lvl - 1)
            if g[lvl - 1] is None:
                g[lvl - 1] = torch.zeros((n, 1, h, w), dtype=torch.float32, device=f1.device)
            check(L.ppv_avgpool2_bwd_acc(ptr(g[lvl]), ptr(g[lvl - 1]), n, h, w, stream_ptr()), "ppv_avgpool2_bwd_acc")
        if g[0] is None:
            return None, None, None
        g1 = torch.empty_like(f1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        check(L.ppv_corr_volume_bwd(ptr(g[0]), ptr(f1), ptr(f2), ptr(g1), ptr(g2), B, C, H * W, stream_ptr()), "ppv_corr_volume_bwd")
        return g1, g2, None


class _LookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, radius, shape, *pyr):
        L = _lib.lib()
        B, H, W = shape
        r = radius
        coords = coords.detach().float().contiguous()
        nl = len(pyr)
        out = torch.empty((B, nl * (2 * r + 1) ** 2, H, W), dtype=torch.float32, device=coords.device)
        import ctypes
        ptrs = (ctypes.c_void_p * nl)(*[c.data_ptr() for c in pyr])
        hs = (ctypes.c_int * nl)(*[c.shape[-2] for c in pyr])
        ws = (ctypes.c_int * nl)(*[c.shape[-1] for c in pyr])
        check(L.ppv_corr_lookup_all(ptrs, hs, ws, nl, ptr(coords), ptr(out), B, H, W, r, stream_ptr()), "ppv_corr_lookup_all")
        ctx.save_for_backward(coords)
        ctx.meta = (r, shape, [tuple(c.shape) for c in pyr])
        return out

    @staticmethod
    def backward(ctx, gout):
        coords, = ctx.saved_tensors
        r, (B, H, W), shapes = ctx.meta
        L = _lib.lib()
        gout = gout.contiguous().float()
        grads = []
        for i, shp in enumerate(shapes):
            gc = torch.zeros(shp, dtype=torch.float32, device=gout.device)
            check(L.ppv_corr_lookup_bwd(ptr(gout), ptr(coords), ptr(gc), B, H, W, shp[-2], shp[-1], r, i, len(shapes), stream_ptr()),
                  "ppv_corr_lookup_bwd")
            grads.append(gc)
        return (None, None, None) + tuple(grads)


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        _lib.require_cuda(fmap1, fmap2)
        self.num_levels, self.radius = num_levels, radius
        B, C, H, W = fmap1.shape
        self.shape = (B, H, W)
        self.corr_pyramid = list(_VolumeFn.apply(fmap1, fmap2, num_levels))

    def __call__(self, coords):
        return _LookupFn.apply(coords, self.radius, self.shape, *self.corr_pyramid)


class _AltCorrFn(torch.autograd.Function):
    """All pyramid levels of the on-the-fly correlation in one launch (f1, f2_l NHWC f32; coords [B,H,W,2] at level 0)."""

    @staticmethod
    def forward(ctx, radius, coords, f1, *f2s):
        import ctypes
        L = _lib.lib()
        B, H1, W1, C = f1.shape
        n = len(f2s)
        rd = 2 * radius + 1
        out = torch.empty((B, n, rd * rd, H1, W1), dtype=torch.float32, device=f1.device)
        scale = 1.0 / float(C) ** 0.5
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in f2s])
        hs = (ctypes.c_int * n)(*[t.shape[1] for t in f2s])
        ws = (ctypes.c_int * n)(*[t.shape[2] for t in f2s])
        check(L.ppv_alt_corr_fwd(ptr(f1), ptrs, hs, ws, n, ptr(coords), ptr(out), B, H1, W1, C, radius, scale, stream_ptr()),
              "ppv_alt_corr_fwd")
        ctx.save_for_backward(f1, coords, *f2s)
        ctx.radius, ctx.scale = radius, scale
        return out

    @staticmethod
    def backward(ctx, g):
        f1, coords, *f2s = ctx.saved_tensors
        L = _lib.lib()
        B, H1, W1, C = f1.shape
        n = len(f2s)
        g = g.contiguous()
        d1 = None
        d2s = []
        for i, f2 in enumerate(f2s):
            want1, want2 = ctx.needs_input_grad[2], ctx.needs_input_grad[3 + i]
            t1 = torch.empty_like(f1) if want1 else None
            t2 = torch.zeros_like(f2) if want2 else None
            if want1 or want2:
                check(L.ppv_alt_corr_bwd(ptr(f1), ptr(f2), ptr(coords), ptr(g), ptr(t1), ptr(t2), B, H1, W1, f2.shape[1], f2.shape[2],
                                         C, ctx.radius, ctx.scale, i, n, stream_ptr()), "ppv_alt_corr_bwd")
            if want1:
                d1 = t1 if d1 is None else d1.add_(t1)
            d2s.append(t2)
        return (None, None, d1) + tuple(d2s)


class AlternateCorrBlock:
    """Drop-in for ``Face-DeId/RAFT/core/corr.py:63 AlternateCorrBlock`` — the memory-efficient correlation that the
    reference backs with its CUDA extension ``alt_cuda_corr`` (and never wires into autograd; here gradients reach both
    feature maps).  Same constructor and ``__call__(coords)``; output equals ``CorrBlock``'s (the pooled-feature and
    pooled-volume pyramids are the same linear map) without ever materialising the HW x HW volume."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        import torch.nn.functional as F
        if not fmap1.is_cuda:
            raise RuntimeError("ppv_amd AlternateCorrBlock runs on an MI355X (fmap1/fmap2 must be cuda tensors); no CPU path")
        self.num_levels, self.radius = num_levels, radius
        self.f1 = fmap1.float().permute(0, 2, 3, 1).contiguous()            # corr.py:82: level-0 fmap1 at every level
        self.f2 = []
        f2 = fmap2.float()
        for i in range(num_levels):
            self.f2.append(f2.permute(0, 2, 3, 1).contiguous())
            f2 = F.avg_pool2d(f2, 2, stride=2)                               # corr.py:72-73

    def __call__(self, coords):
        c = coords.float().permute(0, 2, 3, 1).contiguous()
        B, H, W, _ = c.shape
        return _AltCorrFn.apply(self.radius, c, self.f1, *self.f2).view(B, -1, H, W)
