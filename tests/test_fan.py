"""FAN heat-map regressor: CPU oracle pinned to the golden captured from the reference; HIP drop-in
(bf16 storage, eval-mode BN) against the golden."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from fan_fill import fill_by_name


def _inputs():
    return {"b2_256": torch.rand((2, 3, 256, 256), generator=torch.Generator().manual_seed(0)) * 2 - 1,
            "b1_512": torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(0)) * 2 - 1}


def test_oracle_fan_matches_reference_golden():
    from oracle.fan import FAN
    g = load_golden("fan.npz")
    fan = FAN().eval()
    assert sorted(fan.state_dict().keys()) == list(g["state_names"])
    fill_by_name(fan)
    for tag, x in _inputs().items():
        with torch.no_grad():
            raw = fan(torch.nn.functional.interpolate(x, size=256, mode="bilinear") * 0.5 + 0.5)
            hm = fan.get_heatmap_privacy(x)
        assert rel_err(raw[:, ::7, ::4, ::4], g[f"{tag}_raw_sub"]) < 1e-4
        assert rel_err(hm[0], g[f"{tag}_hm0"]) < 1e-4 and rel_err(hm[1], g[f"{tag}_hm1"]) < 1e-4


@pytest.mark.gpu
def test_hip_fan_matches_reference_golden():
    """Default precision ("fp32": f32 activations, every conv as a [hi | lo | hi] x [W_hi | W_hi | W_lo] bf16 MFMA product with f32
    accumulation): the 99-channel logits and the two clamped 49-channel heat-map sums within the north-star 1e-3 of the
    reference's fp32 outputs, at 256^2 and 512^2 inputs."""
    from ppv_amd.fan import FAN
    g = load_golden("fan.npz")
    fan = FAN().eval()
    assert sorted(fan.state_dict().keys()) == list(g["state_names"])
    fill_by_name(fan)
    fan = fan.cuda()
    for tag, x in _inputs().items():
        hm = fan.get_heatmap(x.cuda(), Privacy=True)
        raw = fan.last_raw
        e_raw = rel_err(raw[:, ::7, ::4, ::4], g[f"{tag}_raw_sub"])
        e0, e1 = rel_err(hm[0], g[f"{tag}_hm0"]), rel_err(hm[1], g[f"{tag}_hm1"])
        print(tag, f"raw {e_raw:.3e} hm0 {e0:.3e} hm1 {e1:.3e}")
        assert hm[0].shape == g[f"{tag}_hm0"].shape and hm[0].dtype == torch.float32
        assert e_raw < 1e-3 and e0 < 1e-3 and e1 < 1e-3


@pytest.mark.gpu
def test_hip_fan_bf16_mode():
    """precision="bf16" (bf16 activation storage through ~45 conv layers, 3x faster): 3e-2 of max on the raw logits; the
    heat-maps are clamped SUMS of 49 such channels on a [0,1] scale: 0.15 absolute plus cosine similarity > 0.99."""
    from ppv_amd.fan import FAN
    g = load_golden("fan.npz")
    fan = FAN(precision="bf16").eval()
    fill_by_name(fan)
    fan = fan.cuda()
    for tag, x in _inputs().items():
        hm = fan.get_heatmap(x.cuda(), Privacy=True)
        e_raw = rel_err(fan.last_raw[:, ::7, ::4, ::4], g[f"{tag}_raw_sub"])
        e0, e1 = rel_err(hm[0], g[f"{tag}_hm0"]), rel_err(hm[1], g[f"{tag}_hm1"])
        assert e_raw < 3e-2 and e0 < 0.15 and e1 < 0.15
        for k, h in enumerate(hm):
            a, b = h.cpu().reshape(-1).double(), torch.tensor(g[f"{tag}_hm{k}"]).reshape(-1).double()
            assert float(a @ b / (a.norm() * b.norm())) > 0.99


@pytest.mark.gpu
def test_get_heatmap_train_carries_the_gradient_to_the_image():
    """wing.py:262-272: the differentiable heat-map path against the reference's own run (tests/golden/fan_train.npz:
    heat-maps and d/d image of a weighted sum of both maps)."""
    from ppv_amd.fan import FAN
    from fan_fill import fill_by_name
    g = load_golden("fan_train.npz")
    fan = FAN().eval()
    fill_by_name(fan)
    fan = fan.cuda()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    hm = fan.get_heatmap_train(x, Privacy=True)
    assert rel_err(hm[0], g["hm0"]) < 1e-3 and rel_err(hm[1], g["hm1"]) < 1e-3
    w0 = torch.rand(hm[0].shape, generator=torch.Generator().manual_seed(6)).cuda()
    w1 = torch.rand(hm[1].shape, generator=torch.Generator().manual_seed(7)).cuda()
    ((hm[0] * w0).sum() + (hm[1] * w1).sum()).backward()
    gx, want = x.grad.cpu().double(), torch.from_numpy(g["gx"]).double()
    rl2, cos = ((gx - want).norm() / want.norm()).item(), float((gx.flatten() @ want.flatten()) / (gx.norm() * want.norm()))
    print(f"get_heatmap_train: d/d image rel L2 {rl2:.2e}, cos {cos:.6f}, max-norm {rel_err(x.grad, g['gx']):.2e}")
    # the heat-maps agree to 1e-3; the gradient passes ~100 ReLU masks and two clamp(0, 1) edges, each of which flips for the few
    # activations that sit within fp32 rounding of its threshold (measured: rel L2 2.3e-2, cos 0.99974, max-norm 5e-2)
    assert rl2 < 5e-2 and cos > 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("Hi,Wi,size,scale,align", [(64, 64, None, 4, True), (200, 180, 256, None, False), (256, 256, 256, None, False),
                                                     (300, 340, 256, None, False), (7, 5, (19, 33), None, True), (17, 9, None, 2, False)])
def test_bilinear_resize_equals_torch_interpolate(Hi, Wi, size, scale, align):
    """nn_ops.bilinear_resize (round 6: the two F.interpolate calls of FAN.get_heatmap_train, wing.py:264,270, on csrc/fan.hip) against
    torch's CPU F.interpolate: forward and the gradient to the input (a deterministic gather), up- and down-sampling, both align_corners
    settings; the train path then patches F.interpolate to raise."""
    import torch.nn.functional as F
    from ppv_amd.nn_ops import bilinear_resize
    g0 = torch.Generator().manual_seed(Hi * 31 + Wi)
    x = torch.randn(2, 3, Hi, Wi, generator=g0)
    xr = x.clone().requires_grad_(True)
    kw = dict(size=size) if size is not None else dict(scale_factor=scale)
    want = F.interpolate(xr, mode="bilinear", align_corners=align, **kw)
    w = torch.randn(want.shape, generator=g0)
    (want * w).sum().backward()
    xd = x.cuda().requires_grad_(True)
    got = bilinear_resize(xd, align_corners=align, **kw)
    assert got.shape == want.shape
    assert rel_err(got, want.detach()) < 1e-5
    (got * w.cuda()).sum().backward()
    assert rel_err(xd.grad, xr.grad) < 1e-5
    again = bilinear_resize(xd, align_corners=align, **kw)
    assert torch.equal(again, got)


@pytest.mark.gpu
def test_get_heatmap_train_calls_no_torch_interpolate(monkeypatch):
    import torch.nn.functional as F
    from ppv_amd.fan import FAN
    from fan_fill import fill_by_name
    fan = FAN().eval()
    fill_by_name(fan)
    fan = fan.cuda()
    x = torch.rand(1, 3, 200, 220, generator=torch.Generator().manual_seed(1)).cuda().requires_grad_(True)

    def boom(*a, **k):
        raise AssertionError("torch F.interpolate called inside FAN.get_heatmap_train")
    monkeypatch.setattr(F, "interpolate", boom)
    hm = fan.get_heatmap_train(x * 2 - 1, Privacy=True)
    (hm[0].sum() + hm[1].sum()).backward()
    assert hm[0].shape == (1, 1, 256, 256) and x.grad is not None and torch.isfinite(x.grad).all()
