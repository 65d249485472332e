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


def alt_corr(fmap1, fmap2, coords, num_levels=4, radius=4):
    """On-the-fly windowed correlation = AlternateCorrBlock (corr.py:63-91) + the semantics of its CUDA extension
    (RAFT/alt_cuda_corr/correlation_kernel.cu:18-119): for every pixel and pyramid level, dot products of fmap1's pixel
    with the (2r+2)^2 integer window of the pooled fmap2 around floor(coords / 2^level) (zero outside the map), each
    scattered with its four bilinear corner weights into the (2r+1)^2 outputs, channel = iy + (2r+1)*ix; / sqrt(C).
    fmap1, fmap2 [B,C,H,W]; coords [B,2,H,W] (x, y).  Mathematically equal to lookup(pyramid(corr_volume(...)))."""
    B, C, H, W = fmap1.shape
    r, rd = radius, 2 * radius + 1
    f1 = fmap1.permute(0, 2, 3, 1)                                                    # [B,H,W,C]
    outs = []
    f2 = fmap2
    for lvl in range(num_levels):
        if lvl > 0:
            f2 = F.avg_pool2d(f2, 2, stride=2)                                         # corr.py:72-73
        H2, W2 = f2.shape[-2:]
        f2n = f2.permute(0, 2, 3, 1)
        c = coords.permute(0, 2, 3, 1) / 2 ** lvl                                      # corr.py:87
        x, y = c[..., 0], c[..., 1]
        x0, y0 = torch.floor(x), torch.floor(y)
        dx, dy = x - x0, y - y0
        out = torch.zeros(B, rd * rd, H, W)
        bi = torch.arange(B).view(B, 1, 1).expand(B, H, W)
        for iy in range(rd + 1):
            for ix in range(rd + 1):
                h2 = y0.long() - r + iy
                w2 = x0.long() - r + ix
                ok = (h2 >= 0) & (h2 < H2) & (w2 >= 0) & (w2 < W2)
                g = f2n[bi, h2.clamp(0, H2 - 1), w2.clamp(0, W2 - 1)]                   # [B,H,W,C]
                s = (f1 * g).sum(-1) * ok
                if iy > 0 and ix > 0:
                    out[:, (iy - 1) + rd * (ix - 1)] += s * dy * dx                    # kernel :101-109 (nw)
                if iy > 0 and ix < rd:
                    out[:, (iy - 1) + rd * ix] += s * dy * (1 - dx)                    # (ne)
                if iy < rd and ix > 0:
                    out[:, iy + rd * (ix - 1)] += s * (1 - dy) * dx                    # (sw)
                if iy < rd and ix < rd:
                    out[:, iy + rd * ix] += s * (1 - dy) * (1 - dx)                    # (se)
        outs.append(out)
    return torch.stack(outs, dim=1).reshape(B, -1, H, W) / torch.sqrt(torch.tensor(float(C)))
