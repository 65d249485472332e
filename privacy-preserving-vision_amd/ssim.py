"""SSIM loss on the MI355X — drop-in for ``Image_Caption/pytorch_ssim`` (``SSIM`` module and ``ssim`` function, window 11,
sigma 1.5; ``camera_loss = 'SSIM'`` in train.py:172-173).  One fused kernel per direction (csrc/ssim.hip): the separable
Gaussian moments, the SSIM map and its mean never leave the workgroup; gradients flow to both images.  No CPU path."""
import ctypes
import math

import torch
from torch import nn

from . import _lib
from ._lib import check, ptr, stream_ptr


def _taps(window_size, sigma=1.5):
    g = torch.tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(window_size)])
    g = (g / g.sum()).tolist()                                     # pytorch_ssim/__init__.py:8-10, float32 like the reference
    return (ctypes.c_float * 11)(*g)


class _SsimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, size_average):
        if not (img1.is_cuda and img2.is_cuda):
            raise RuntimeError("ppv_amd ssim runs on an MI355X (cuda tensors); no CPU path")
        a, b = img1.contiguous().float(), img2.contiguous().float()
        B, C, H, W = a.shape
        taps = _taps(11)
        sums = torch.zeros(B, dtype=torch.float64, device=a.device)
        check(_lib.lib().ppv_ssim_fwd(ptr(a), ptr(b), ptr(sums), taps, B, C, H, W, stream_ptr()), "ppv_ssim_fwd")
        ctx.save_for_backward(a, b)
        ctx.size_average, ctx.taps = size_average, taps
        per = (sums / (C * H * W)).float()
        return per.mean() if size_average else per

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        B, C, H, W = a.shape
        if ctx.size_average:
            gs = (g.float() / (B * C * H * W)).expand(B).contiguous()
        else:
            gs = (g.float() / (C * H * W)).contiguous()
        d1 = d2 = None
        L = _lib.lib()
        if ctx.needs_input_grad[1]:
            d2 = torch.empty_like(b)
            check(L.ppv_ssim_bwd(ptr(a), ptr(b), ptr(gs), ptr(d2), ctx.taps, B, C, H, W, stream_ptr()), "ppv_ssim_bwd")
        if ctx.needs_input_grad[0]:                                # the map is symmetric in its two images
            d1 = torch.empty_like(a)
            check(L.ppv_ssim_bwd(ptr(b), ptr(a), ptr(gs), ptr(d1), ctx.taps, B, C, H, W, stream_ptr()), "ppv_ssim_bwd")
        return d1, d2, None


def ssim(img1, img2, window_size=11, size_average=True):
    if window_size != 11:
        raise NotImplementedError("ppv_amd ssim: the reference only ever uses window_size = 11 (pytorch_ssim/__init__.py:43,66)")
    return _SsimFn.apply(img1, img2, size_average)


class SSIM(nn.Module):
    """pytorch_ssim/__init__.py:43-64."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size, self.size_average = window_size, size_average

    def forward(self, img1, img2):
        return ssim(img1, img2, self.window_size, self.size_average)
