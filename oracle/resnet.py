"""CPU oracle (torch.nn, fp32) for the reference Encoder: torchvision ResNet-101 trunk + AdaptiveAvgPool2d + NHWC.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates reference Image_Caption/models.py:8-54 (Encoder) with the
public torchvision ResNet-101 architecture (v1.5 Bottleneck: stride on the 3x3; layers [3,4,23,3]).
torchvision itself is absent here; PINNED (round 3) against tests/golden/encoder.npz = the reference's own models.Encoder class run
on CPU over a stand-in ``torchvision.models.resnet101`` assembled from this file's pieces (tests/test_oracle_encoder_golden.py), and
cross-checked against the independent ResNet-101 v1.5 of ``transformers`` (tests/test_oracle_trunk_pin.py).
``round_bf16=True`` rounds the conv outputs and the activations to bfloat16 at the same points the MI355X trunk
stores them, so the comparison isolates kernel arithmetic from the storage format.
"""
import torch
import torch.nn.functional as F
from torch import nn


def _r(t, on):
    return t.bfloat16().float() if on else t


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _rs(t, on):
    return _RoundSTE.apply(t) if on else t


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample
        self.stride = stride
        self.round_bf16 = False

    def _conv(self, conv, x):
        w = _r(conv.weight, self.round_bf16) if not conv.weight.requires_grad else _rs(conv.weight, self.round_bf16)
        return _rs(F.conv2d(x, w, stride=conv.stride, padding=conv.padding), self.round_bf16)

    def forward(self, x):
        rb = self.round_bf16
        out = _rs(self.relu(self.bn1(self._conv(self.conv1, x))), rb)
        out = _rs(self.relu(self.bn2(self._conv(self.conv2, out))), rb)
        out = self.bn3(self._conv(self.conv3, out))
        identity = x
        if self.downsample is not None:
            identity = self.downsample[1](self._conv(self.downsample[0], x))
        return _rs(self.relu(out + identity), rb)


def make_resnet101_trunk(layers=(3, 4, 23, 3)):
    """children()[:-2] of torchvision.models.resnet101 as an nn.Sequential (models.py:17-21)."""
    inplanes = 64
    mods = [nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=False),
            nn.MaxPool2d(3, stride=2, padding=1)]
    for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
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
    for m in net.modules():                      # torchvision's init
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return net


class Encoder(nn.Module):
    """models.py:8-54."""

    def __init__(self, encoded_image_size=36, layers=(3, 4, 23, 3), round_bf16=False):
        super().__init__()
        self.enc_image_size = encoded_image_size
        self.resnet = make_resnet101_trunk(layers)
        self.adaptive_pool = nn.AdaptiveAvgPool2d((encoded_image_size, encoded_image_size))
        self.round_bf16 = round_bf16
        for m in self.resnet.modules():
            if isinstance(m, Bottleneck):
                m.round_bf16 = round_bf16
        self.fine_tune()

    def forward(self, images):
        rb = self.round_bf16
        r = self.resnet
        x = _rs(images, rb)
        x = _rs(F.conv2d(x, _r(r[0].weight, rb), stride=2, padding=3), rb)
        x = _rs(r[2](r[1](x)), rb)
        x = r[3](x)
        for i in range(4, 8):
            x = r[i](x)
        out = self.adaptive_pool(x)
        return out.permute(0, 2, 3, 1)

    def fine_tune(self, fine_tune=True):
        for p in self.resnet.parameters():
            p.requires_grad = False
        for c in list(self.resnet.children())[5:]:
            for p in c.parameters():
                p.requires_grad = fine_tune
