"""CPU oracle (torch.nn, fp32) for the FAN heat-map regressor used by Face-DeId (forward / get_heatmap).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates reference Face-DeId/core/wing.py:36-75 (HourGlass), :78-136
(AddCoordsTh / CoordConvTh), :139-175 (ConvBlock), :178-260 (FAN.forward, get_heatmap) with identical state_dict
names, so weights can be filled by name (tests/golden/make_golden.py::fill_by_name)."""
import torch
import torch.nn.functional as F
from torch import nn


def coord_channels(h, w, with_r=True):
    """wing.py:86-99: xx varies along H, yy along W, both in [-1, 1]; rr = radius / max radius."""
    xx = (torch.arange(h).unsqueeze(1).expand(h, w).float() / (h - 1)) * 2 - 1
    yy = (torch.arange(w).unsqueeze(0).expand(h, w).float() / (w - 1)) * 2 - 1
    ch = [xx, yy]
    if with_r:
        rr = torch.sqrt(xx ** 2 + yy ** 2)
        ch.append(rr / rr.max())
    return torch.stack(ch, 0).unsqueeze(0)


class CoordConv(nn.Module):
    def __init__(self, h, w, cin, cout, **kw):
        super().__init__()
        self.register_buffer("coords", coord_channels(h, w), persistent=False)
        self.conv = nn.Conv2d(cin + 3, cout, **kw)

    def forward(self, x):
        return self.conv(torch.cat([x, self.coords.expand(x.shape[0], -1, -1, -1)], 1))


class ConvBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(cin)
        self.conv1 = nn.Conv2d(cin, cout // 2, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout // 2)
        self.conv2 = nn.Conv2d(cout // 2, cout // 4, 3, 1, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout // 4)
        self.conv3 = nn.Conv2d(cout // 4, cout // 4, 3, 1, 1, bias=False)
        self.downsample = None
        if cin != cout:
            self.downsample = nn.Sequential(nn.BatchNorm2d(cin), nn.ReLU(False), nn.Conv2d(cin, cout, 1, 1, bias=False))

    def forward(self, x):
        o1 = self.conv1(F.relu(self.bn1(x)))
        o2 = self.conv2(F.relu(self.bn2(o1)))
        o3 = self.conv3(F.relu(self.bn3(o2)))
        res = x if self.downsample is None else self.downsample(x)
        return torch.cat((o1, o2, o3), 1) + res


class HourGlass(nn.Module):
    def __init__(self, depth=4):
        super().__init__()
        self.depth = depth
        self.coordconv = CoordConv(64, 64, 256, 256, kernel_size=1, stride=1, padding=0)
        for lvl in range(depth, 0, -1):
            self.add_module(f"b1_{lvl}", ConvBlock(256, 256))
            self.add_module(f"b2_{lvl}", ConvBlock(256, 256))
            if lvl == 1:
                self.add_module("b2_plus_1", ConvBlock(256, 256))
        for lvl in range(1, depth + 1):
            self.add_module(f"b3_{lvl}", ConvBlock(256, 256))

    def _fwd(self, lvl, x):
        up1 = self._modules[f"b1_{lvl}"](x)
        low = self._modules[f"b2_{lvl}"](F.avg_pool2d(x, 2, stride=2))
        low = self._fwd(lvl - 1, low) if lvl > 1 else self._modules["b2_plus_1"](low)
        low = self._modules[f"b3_{lvl}"](low)
        return up1 + F.interpolate(low, scale_factor=2, mode="nearest")

    def forward(self, x):
        return self._fwd(self.depth, self.coordconv(x))


class FAN(nn.Module):
    def __init__(self, num_landmarks=98):
        super().__init__()
        self.conv1 = CoordConv(256, 256, 3, 64, kernel_size=7, stride=2, padding=3)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = ConvBlock(64, 128)
        self.conv3 = ConvBlock(128, 128)
        self.conv4 = ConvBlock(128, 256)
        self.m0 = HourGlass(4)
        self.top_m_0 = ConvBlock(256, 256)
        self.conv_last0 = nn.Conv2d(256, 256, 1, 1, 0)
        self.bn_end0 = nn.BatchNorm2d(256)
        self.l0 = nn.Conv2d(256, num_landmarks + 1, 1, 1, 0)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.avg_pool2d(self.conv2(x), 2, stride=2)
        x = self.conv4(self.conv3(x))
        ll = self.top_m_0(self.m0(x))
        ll = F.relu(self.bn_end0(self.conv_last0(ll)))
        return self.l0(ll)

    @torch.no_grad()
    def get_heatmap_privacy(self, x):
        """wing.py:241-251 with b_preprocess=True, Privacy=True."""
        x = F.interpolate(x, size=256, mode="bilinear")
        hm = self(x * 0.5 + 0.5)[:, :-1]
        hm = F.interpolate(hm, scale_factor=x.size(2) // hm.size(2), mode="bilinear", align_corners=True)
        return [hm[:, :49].sum(dim=1, keepdim=True).clamp_(0, 1), hm[:, 49:].sum(dim=1, keepdim=True).clamp_(0, 1)]
