"""Drop-in for reference ``Face-DeId/RAFT/core/corr.py:12 CorrBlock`` (forward) on MI355X: all-pairs correlation
volume (fp32 MFMA), 4-level average-pool pyramid and the 9x9 bilinear window lookup, through libppv_hip.so.
Same constructor and ``__call__(coords)`` contract; no autograd graph is attached this round (gap: RAFT's flow loss
back-propagates through the lookup in ``core/utils.py:437-462``)."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        _lib.require_cuda(fmap1, fmap2)
        self.num_levels, self.radius = num_levels, radius
        f1 = fmap1.detach().float().contiguous()
        f2 = fmap2.detach().float().contiguous()
        B, C, H, W = f1.shape
        self.shape = (B, H, W)
        L = _lib.lib()
        corr = torch.empty((B * H * W, 1, H, W), dtype=torch.float32, device=f1.device)
        check(L.ppv_corr_volume(ptr(f1), ptr(f2), ptr(corr), B, C, H * W, stream_ptr()), "ppv_corr_volume")
        self.corr_pyramid = [corr]
        h, w = H, W
        for _ in range(num_levels - 1):
            nxt = torch.empty((B * H * W, 1, h // 2, w // 2), dtype=torch.float32, device=f1.device)
            check(L.ppv_avgpool2(ptr(self.corr_pyramid[-1]), ptr(nxt), B * H * W, h, w, stream_ptr()), "ppv_avgpool2")
            self.corr_pyramid.append(nxt)
            h, w = h // 2, w // 2

    def __call__(self, coords):
        B, H, W = self.shape
        r = self.radius
        coords = coords.detach().float().contiguous()
        out = torch.empty((B, self.num_levels * (2 * r + 1) ** 2, H, W), dtype=torch.float32, device=coords.device)
        L = _lib.lib()
        for i, c in enumerate(self.corr_pyramid):
            check(L.ppv_corr_lookup(ptr(c), ptr(coords), ptr(out), B, H, W, c.shape[-2], c.shape[-1], r, i, self.num_levels,
                                    stream_ptr()), "ppv_corr_lookup")
        return out
