"""CPU oracle (torch-CPU ops) for RAFT's CorrBlock.  TEST INFRASTRUCTURE (see oracle/__init__.py).
Restates reference Face-DeId/RAFT/core/corr.py:12-60 and RAFT/core/utils/utils.py:57-71 (bilinear_sampler)."""
import torch
import torch.nn.functional as F


def corr_volume(fmap1, fmap2):
    b, c, h, w = fmap1.shape
    corr = torch.matmul(fmap1.view(b, c, h * w).transpose(1, 2), fmap2.view(b, c, h * w))
    return corr.view(b, h, w, 1, h, w) / torch.sqrt(torch.tensor(c).float())          # corr.py:53-60


def pyramid(corr, num_levels=4):
    b, h1, w1, d, h2, w2 = corr.shape
    corr = corr.reshape(b * h1 * w1, d, h2, w2)
    out = [corr]
    for _ in range(num_levels - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)                                       # corr.py:25-27
        out.append(corr)
    return out


def lookup(pyr, coords, radius=4):
    r = radius
    coords = coords.permute(0, 2, 3, 1)
    b, h1, w1, _ = coords.shape
    outs = []
    for i, corr in enumerate(pyr):
        dx = torch.linspace(-r, r, 2 * r + 1)
        dy = torch.linspace(-r, r, 2 * r + 1)
        delta = torch.stack(torch.meshgrid(dy, dx, indexing="ij"), dim=-1)             # corr.py:37-39
        cl = coords.reshape(b * h1 * w1, 1, 1, 2) / 2 ** i + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        hh, ww = corr.shape[-2:]
        xg, yg = cl.split([1, 1], dim=-1)
        grid = torch.cat([2 * xg / (ww - 1) - 1, 2 * yg / (hh - 1) - 1], dim=-1)      # utils.py:59-64
        s = F.grid_sample(corr, grid, align_corners=True)
        outs.append(s.view(b, h1, w1, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().float()
