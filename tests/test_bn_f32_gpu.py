"""csrc/bn_f32.hip (the element-wise side of Encoder(precision="fp32")) against stock PyTorch on the CPU in f64: train / eval BatchNorm2d
(+ residual) (+ ReLU) forward, backward and running statistics; MaxPool2d(3, 2, 1) incl. the first-maximum rule; AdaptiveAvgPool2d; the
six-part operand split.  Reference semantics: torchvision ResNet-101 layers behind Image_Caption/models.py:17-41."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("res,relu", [(False, True), (True, True), (True, False), (False, False)])
@pytest.mark.parametrize("shape", [(4, 64, 9, 7), (3, 256, 16, 16), (37, 128, 5, 5), (4, 256, 8, 8), (4, 2048, 2, 2)])
def test_batch_norm_f32_matches_torch(shape, res, relu, train):
    from ppv_amd.nn_ops import batch_norm_f32
    B, C, H, W = shape
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(shape, generator=g) * 1.7 + torch.randn(1, C, 1, 1, generator=g)).double()
    r = torch.randn(shape, generator=g).double() if res else None
    gy = torch.randn(shape, generator=g).double()
    bn_ref = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        bn_ref.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn_ref.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn_ref.running_mean.copy_(torch.randn(C, generator=g) * 0.2)
        bn_ref.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn = torch.nn.BatchNorm2d(C)
    bn.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in bn_ref.state_dict().items()})
    bn = bn.cuda()
    bn_ref.train(train)
    bn.train(train)
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    t = bn_ref(xr)
    if res:
        t = t + rr
    yr = torch.relu(t) if relu else t
    yr.backward(gy)
    xg = _nhwc(x.float()).cuda().requires_grad_(True)
    rg = _nhwc(r.float()).cuda().requires_grad_(True) if res else None
    y = batch_norm_f32(xg, bn, res=rg, relu=relu)
    y.backward(_nhwc(gy.float()).cuda())
    tol = dict(rtol=2e-5, atol=2e-5)
    assert torch.allclose(y.detach().cpu().double(), _nhwc(yr.detach()), **tol)
    scale = _nhwc(xr.grad).abs().max().item()
    assert (xg.grad.cpu().double() - _nhwc(xr.grad)).abs().max().item() < 5e-5 * max(scale, 1.0)
    if res:
        assert torch.allclose(rg.grad.cpu().double(), _nhwc(rr.grad), **tol)
    assert torch.allclose(bn.weight.grad.cpu().double(), bn_ref.weight.grad, rtol=1e-4, atol=1e-4 * bn_ref.weight.grad.abs().max().item())
    assert torch.allclose(bn.bias.grad.cpu().double(), bn_ref.bias.grad, rtol=1e-4, atol=1e-4 * bn_ref.bias.grad.abs().max().item())
    assert torch.allclose(bn.running_mean.cpu().double(), bn_ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu().double(), bn_ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(bn_ref.num_batches_tracked)
    # deterministic: the same bits again
    bn2 = torch.nn.BatchNorm2d(C).cuda()
    bn2.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in bn_ref.state_dict().items()})
    bn2.train(train)
    assert torch.equal(batch_norm_f32(xg.detach(), bn2, res=None if rg is None else rg.detach(), relu=relu),
                       batch_norm_f32(xg.detach(), bn2, res=None if rg is None else rg.detach(), relu=relu)) or train


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (3, 64, 9, 11), (2, 8, 7, 7)])
def test_max_pool_f32_matches_torch_incl_the_first_maximum_rule(shape):
    from ppv_amd.nn_ops import max_pool3x3s2_f32
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(shape, generator=g))               # zeros in runs: ties everywhere
    x = (x * 4).round() / 4                                       # coarse grid: ties among the positive values too
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = _nhwc(x).cuda().requires_grad_(True)
    y = max_pool3x3s2_f32(xg)
    y.backward(_nhwc(gy).cuda())
    assert torch.equal(y.detach().cpu(), _nhwc(yr.detach()))
    assert torch.allclose(xg.grad.cpu(), _nhwc(xr.grad), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("hw,E", [(8, 36), (8, 3), (7, 5), (2, 36), (16, 16)])
def test_adaptive_avg_pool_f32_matches_torch(hw, E):
    from ppv_amd.nn_ops import adaptive_avg_pool_f32
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 64, hw, hw, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.adaptive_avg_pool2d(xr, E)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = _nhwc(x).cuda().requires_grad_(True)
    y = adaptive_avg_pool_f32(xg, E)
    y.backward(_nhwc(gy).cuda())
    assert torch.allclose(y.detach().cpu(), _nhwc(yr.detach()), rtol=1e-5, atol=1e-6)
    assert torch.allclose(xg.grad.cpu(), _nhwc(xr.grad), rtol=1e-5, atol=1e-5)


def test_split6_reconstructs_the_operand_to_f32_level():
    from ppv_amd.nn_ops import _split6
    x = torch.randn(5, 7, 7, 72, generator=torch.Generator().manual_seed(3)) * 3
    y = _split6(x.cuda()).float().cpu()
    C = 72
    assert y.shape[-1] == 448 and not y[..., 6 * C:].any()
    h, m, l = y[..., :C], y[..., C:2 * C], y[..., 3 * C:4 * C]
    assert torch.equal(h, y[..., 2 * C:3 * C]) and torch.equal(h, y[..., 4 * C:5 * C]) and torch.equal(m, y[..., 5 * C:6 * C])
    assert torch.equal(h, x.bfloat16().float())
    assert ((h.double() + m.double() + l.double() - x.double()).abs() <= 2.0 ** -24 * x.abs().double() + 1e-30).all()
