"""SSIM loss: oracle vs the reference's own outputs on CPU; fused HIP kernels vs the same golden."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_matches_reference_golden(tag):
    from oracle import ssim as oss
    g = load_golden("ssim.npz")
    x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
    y = torch.from_numpy(g[f"{tag}_y"]).requires_grad_(True)
    v = oss.ssim(x, y)
    assert abs(v.item() - float(g[f"{tag}_ssim"])) < 1e-6
    v.backward()
    assert rel_err(x.grad, g[f"{tag}_gx"]) < 1e-5 and rel_err(y.grad, g[f"{tag}_gy"]) < 1e-5
    assert rel_err(oss.ssim(x.detach(), y.detach(), size_average=False), g[f"{tag}_per_image"]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_ssim_matches_reference_golden(tag):
    import ppv_amd.ssim as ps
    g = load_golden("ssim.npz")
    x = torch.from_numpy(g[f"{tag}_x"]).cuda().requires_grad_(True)
    y = torch.from_numpy(g[f"{tag}_y"]).cuda().requires_grad_(True)
    v = ps.SSIM()(x, y)
    assert abs(v.item() - float(g[f"{tag}_ssim"])) < 1e-5
    v.backward()
    assert rel_err(x.grad, g[f"{tag}_gx"]) < 1e-3 and rel_err(y.grad, g[f"{tag}_gy"]) < 1e-3       # north_star: 1e-3 rel fp32
    per = ps.ssim(x.detach(), y.detach(), size_average=False)
    assert rel_err(per, g[f"{tag}_per_image"]) < 1e-5
    # per-image weights reach the right images
    w = torch.tensor(np.arange(1, per.numel() + 1, dtype=np.float32)).cuda()
    y2 = y.detach().clone().requires_grad_(True)
    (ps.ssim(x.detach(), y2, size_average=False) * w).sum().backward()
    B = per.numel()
    want = torch.from_numpy(g[f"{tag}_gy"]).cuda() * B * w.view(B, 1, 1, 1)
    assert rel_err(y2.grad, want) < 1e-3


@pytest.mark.gpu
def test_hip_ssim_full_size_properties():
    import ppv_amd.ssim as ps
    x = torch.rand(128, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    assert abs(ps.ssim(x, x).item() - 1.0) < 1e-5                                 # SSIM(x, x) = 1
    y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
    a, b = ps.ssim(x, y).item(), ps.ssim(y, x).item()
    assert abs(a - b) < 1e-6 and 0.0 < a < 1.0                                     # symmetric
