"""CPU oracle (torch fp32) for the StarGAN-v2 blocks.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates reference Face-DeId/core/model.py:12-124 functionally (ResBlk :12-53,
AdaIN :56-66, AdainResBlk :69-110) over a dict of parameters under the reference's names; pinned by tests/golden/stargan.npz
(the reference modules themselves, tests/golden/make_golden.py stargan)."""
import math

import torch.nn.functional as F


def res_blk(x, p, normalize, downsample, slope=0.2):
    sc = F.conv2d(x, p["conv1x1.weight"]) if "conv1x1.weight" in p else x
    if downsample:
        sc = F.avg_pool2d(sc, 2)
    r = x
    if normalize:
        r = F.instance_norm(r, weight=p["norm1.weight"], bias=p["norm1.bias"], eps=1e-5)
    r = F.conv2d(F.leaky_relu(r, slope), p["conv1.weight"], p["conv1.bias"], padding=1)
    if downsample:
        r = F.avg_pool2d(r, 2)
    if normalize:
        r = F.instance_norm(r, weight=p["norm2.weight"], bias=p["norm2.bias"], eps=1e-5)
    r = F.conv2d(F.leaky_relu(r, slope), p["conv2.weight"], p["conv2.bias"], padding=1)
    return (sc + r) / math.sqrt(2)


def adain(x, s, w, b):
    h = F.linear(s, w, b)
    gamma, beta = h.chunk(2, dim=1)
    return (1 + gamma[:, :, None, None]) * F.instance_norm(x, eps=1e-5) + beta[:, :, None, None]


def adain_res_blk(x, s, p, upsample, w_hpf=0, slope=0.2):
    r = F.leaky_relu(adain(x, s, p["norm1.fc.weight"], p["norm1.fc.bias"]), slope)
    if upsample:
        r = F.interpolate(r, scale_factor=2, mode="nearest")
    r = F.conv2d(r, p["conv1.weight"], p["conv1.bias"], padding=1)
    r = F.conv2d(F.leaky_relu(adain(r, s, p["norm2.fc.weight"], p["norm2.fc.bias"]), slope), p["conv2.weight"], p["conv2.bias"], padding=1)
    if w_hpf == 0:
        sc = F.interpolate(x, scale_factor=2, mode="nearest") if upsample else x
        if "conv1x1.weight" in p:
            sc = F.conv2d(sc, p["conv1x1.weight"])
        r = (r + sc) / math.sqrt(2)
    return r
