"""CPU oracle (TEST INFRASTRUCTURE ONLY) of the reference's SSIM loss, Image_Caption/pytorch_ssim/__init__.py:8-40:
normalised 1-D Gaussian (11 taps, sigma 1.5), its outer product as the depthwise window, zero padding, C1 = 0.01^2,
C2 = 0.03^2.  Pinned by tests/golden/ssim.npz (the reference module itself run on CPU: value, per-image values and both
image gradients)."""
import math
import torch
import torch.nn.functional as F


def gaussian_taps(window_size=11, sigma=1.5):
    g = torch.tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(window_size)])   # :8-10
    return g / g.sum()


def ssim_map(img1, img2, window_size=11):
    C = img1.shape[1]
    t = gaussian_taps(window_size).unsqueeze(1)
    win = (t @ t.t()).float()[None, None].expand(C, 1, window_size, window_size).contiguous()                      # :13-17
    blur = lambda z: F.conv2d(z, win, padding=window_size // 2, groups=C)
    mu1, mu2 = blur(img1), blur(img2)
    s11, s22, s12 = blur(img1 * img1) - mu1 * mu1, blur(img2 * img2) - mu2 * mu2, blur(img1 * img2) - mu1 * mu2     # :20-30
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))              # :32-35


def ssim(img1, img2, window_size=11, size_average=True):
    m = ssim_map(img1, img2, window_size)
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)                                                # :37-40
