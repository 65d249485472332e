"""Drop-ins for the StarGAN-v2 building blocks of reference ``Face-DeId/core/model.py:12-124``: ``ResBlk``, ``AdaIN``,
``AdainResBlk`` -- same constructors, parameter names (``conv1 / conv2 / conv1x1 / norm1 / norm2 / fc``) and NCHW ``forward``;
the convolutions and the InstanceNorm / AdaIN (+ LeakyReLU) run on the HIP kernels (ppv_amd.nn_ops), forward and backward."""
import math

import torch
from torch import nn

from .nn_ops import conv2d_f32, instance_norm_act


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().float()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _avgpool2(x):                                   # F.avg_pool2d(x, 2) on NHWC
    B, H, W, C = x.shape
    return x.view(B, H // 2, 2, W // 2, 2, C).mean(dim=(2, 4))


def _up2(x):                                        # F.interpolate(scale_factor=2, mode='nearest') on NHWC
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def _slope(actv):
    if isinstance(actv, nn.LeakyReLU):
        return actv.negative_slope
    if isinstance(actv, nn.ReLU):
        return 0.0
    return None


class ResBlk(nn.Module):
    def __init__(self, dim_in, dim_out, actv=nn.LeakyReLU(0.2), normalize=False, downsample=False):
        super().__init__()
        self.actv = actv
        self.normalize = normalize
        self.downsample = downsample
        self.learned_sc = dim_in != dim_out
        self.conv1 = nn.Conv2d(dim_in, dim_in, 3, 1, 1)
        self.conv2 = nn.Conv2d(dim_in, dim_out, 3, 1, 1)
        if self.normalize:
            self.norm1 = nn.InstanceNorm2d(dim_in, affine=True)
            self.norm2 = nn.InstanceNorm2d(dim_in, affine=True)
        if self.learned_sc:
            self.conv1x1 = nn.Conv2d(dim_in, dim_out, 1, 1, 0, bias=False)

    def _norm_act(self, x, norm):
        s = _slope(self.actv)
        if self.normalize:
            if s is not None:
                return instance_norm_act(x, norm.weight, norm.bias, s, norm.eps)
            x = instance_norm_act(x, norm.weight, norm.bias, 1.0, norm.eps)
        return self.actv(x)

    def forward(self, x):
        x = _nhwc(x)
        sc = conv2d_f32(x, self.conv1x1.weight) if self.learned_sc else x
        if self.downsample:
            sc = _avgpool2(sc)
        r = conv2d_f32(self._norm_act(x, getattr(self, "norm1", None)), self.conv1.weight, self.conv1.bias, 1, 1)
        if self.downsample:
            r = _avgpool2(r)
        r = conv2d_f32(self._norm_act(r, getattr(self, "norm2", None)), self.conv2.weight, self.conv2.bias, 1, 1)
        return _nchw((sc + r) / math.sqrt(2))


class AdaIN(nn.Module):
    def __init__(self, style_dim, num_features):
        super().__init__()
        self.norm = nn.InstanceNorm2d(num_features, affine=False)
        self.fc = nn.Linear(style_dim, num_features * 2)

    def nhwc(self, x, s, slope=1.0):
        h = self.fc(s)
        gamma, beta = torch.chunk(h, chunks=2, dim=1)
        return instance_norm_act(x, 1 + gamma, beta, slope, self.norm.eps)

    def forward(self, x, s):
        return _nchw(self.nhwc(_nhwc(x), s))


class AdainResBlk(nn.Module):
    def __init__(self, dim_in, dim_out, style_dim=64, w_hpf=0, actv=nn.LeakyReLU(0.2), upsample=False):
        super().__init__()
        self.w_hpf = w_hpf
        self.actv = actv
        self.upsample = upsample
        self.learned_sc = dim_in != dim_out
        self.conv1 = nn.Conv2d(dim_in, dim_out, 3, 1, 1)
        self.conv2 = nn.Conv2d(dim_out, dim_out, 3, 1, 1)
        self.norm1 = AdaIN(style_dim, dim_in)
        self.norm2 = AdaIN(style_dim, dim_out)
        if self.learned_sc:
            self.conv1x1 = nn.Conv2d(dim_in, dim_out, 1, 1, 0, bias=False)

    def _norm_act(self, x, norm, s):
        sl = _slope(self.actv)
        if sl is not None:
            return norm.nhwc(x, s, sl)
        return self.actv(norm.nhwc(x, s))

    def forward(self, x, s):
        x = _nhwc(x)
        r = self._norm_act(x, self.norm1, s)
        if self.upsample:
            r = _up2(r)
        r = conv2d_f32(r, self.conv1.weight, self.conv1.bias, 1, 1)
        r = conv2d_f32(self._norm_act(r, self.norm2, s), self.conv2.weight, self.conv2.bias, 1, 1)
        if self.w_hpf == 0:
            sc = _up2(x) if self.upsample else x
            if self.learned_sc:
                sc = conv2d_f32(sc, self.conv1x1.weight)
            r = (r + sc) / math.sqrt(2)
        return _nchw(r)
