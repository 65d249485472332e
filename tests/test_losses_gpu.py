"""ppv_amd.losses (fused camera MSE term of Image_Caption/train.py:170-171,284-288) against stock PyTorch on the CPU in f64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g), torch.rand(shape, generator=g) * 0.8


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (3, 3, 37, 41), (1, 1, 1, 3), (16, 3, 256, 256)])
def test_mse_value_and_gradients_equal_torch(shape):
    from ppv_amd.losses import mse_loss, one_minus_mse
    a, b = _pair(shape, 0)
    ad, bd = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(ad, bd)
    (3.0 * (1 - ref)).backward()
    ag, bg = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    got = one_minus_mse(ag, bg)
    (3.0 * got).backward()
    assert abs(got.item() - (1 - ref.item())) < 2e-7 * max(1.0, abs(1 - ref.item()))
    assert torch.allclose(bg.grad.cpu().double(), bd.grad, rtol=1e-5, atol=1e-12)
    assert torch.allclose(ag.grad.cpu().double(), ad.grad, rtol=1e-5, atol=1e-12)
    m = mse_loss(a.cuda(), b.cuda())
    assert abs(m.item() - ref.item()) < 2e-7
    # deterministic: same bits on a second evaluation
    assert one_minus_mse(ag, bg).item() == got.item()


def test_tap_adds_the_other_consumers_gradient_in_the_same_pass():
    """camera_mse_tap(imgs, sensor): the returned tensor IS sensor (same storage); the gradient of a second consumer of it and the loss'
    own gradient arrive at `sensor` summed, as autograd's accumulation would give with the torch expression."""
    from ppv_amd.losses import camera_mse_tap
    imgs, sensor0 = _pair((4, 3, 64, 64), 1)
    w = torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(2))
    sd = sensor0.double().requires_grad_(True)
    ref = 6 * (1 - torch.nn.functional.mse_loss(imgs.double(), sd)) + 0.4 * (sd * w.double()).sum()
    ref.backward()
    sg = sensor0.cuda().requires_grad_(True)
    s_leaf = sg * 1.0                                            # a non-leaf, like the camera's output
    tapped, loss_cam = camera_mse_tap(imgs.cuda(), s_leaf)
    assert tapped.data_ptr() == s_leaf.data_ptr()
    total = 6 * loss_cam + 0.4 * (tapped * w.cuda()).sum()
    total.backward()
    assert abs(total.item() - ref.item()) < 1e-3 * abs(ref.item())
    assert torch.allclose(sg.grad.cpu().double(), sd.grad, rtol=1e-5, atol=1e-9)
    # loss term unused: the other consumer's gradient passes through untouched
    sg2 = sensor0.cuda().requires_grad_(True)
    tapped2, _ = camera_mse_tap(imgs.cuda(), sg2 * 1.0)
    (tapped2 * w.cuda()).sum().backward()
    assert torch.allclose(sg2.grad.cpu(), w, rtol=0, atol=0)
    # encoder unused: only the loss' gradient
    sg3 = sensor0.cuda().requires_grad_(True)
    _, l3 = camera_mse_tap(imgs.cuda(), sg3 * 1.0)
    l3.backward()
    sd3 = sensor0.double().requires_grad_(True)
    (1 - torch.nn.functional.mse_loss(imgs.double(), sd3)).backward()
    assert torch.allclose(sg3.grad.cpu().double(), sd3.grad, rtol=1e-5, atol=1e-12)
