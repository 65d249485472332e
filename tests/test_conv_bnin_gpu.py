"""conv2 of an identity bottleneck with bn1 + ReLU applied inside the convolution (csrc/conv_halo.hip BNIN, round 6) against the
two-launch form it replaces (ppv_bn_act_fold_rows + ppv_conv_gemm: bit for bit -- both derive the coefficients through csrc/bn_coef.h)
and against stock PyTorch on the CPU (train-mode BatchNorm2d + ReLU + Conv2d of torchvision's Bottleneck, Image_Caption/models.py:17-21
under train.py:245)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _case(B, H, C, seed):
    import ppv_amd.convops as co
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, H, H, C, generator=g) * 1.3 + 0.2 * torch.randn(C, generator=g)).to(BF16).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    bn = torch.nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    xf = x.float().view(-1, C)
    half = xf.shape[0] // 2
    sums = torch.stack([torch.stack([xf[:half].sum(0), (xf[:half] ** 2).sum(0)]), torch.stack([xf[half:].sum(0), (xf[half:] ** 2).sum(0)])]).contiguous()
    return co, x, w, bn, sums


@pytest.mark.parametrize("B,H,C", [(128, 16, 256), (64, 32, 128)])
def test_bnin_equals_the_two_launch_form_bit_for_bit(B, H, C):
    import copy
    co, x, w, bn, sums = _case(B, H, C, 0)
    assert co.conv3x3_bnin_supported(B, H, H, C, C)
    wt = co.weight_layout(w, 0)
    M = B * H * H
    rows = co.stat_tiles(M)
    bn_a, bn_b = copy.deepcopy(bn), copy.deepcopy(bn)
    # the form it replaces
    y_ref, _, coef_ref = co.bn_act_fold(x, sums, M, bn_a, 0.1)
    st_ref = torch.zeros(rows, 2, C, device="cuda")
    out_ref = co.conv_fwd(y_ref, wt, 1, 1, stat_part=st_ref)
    # one launch
    st = torch.zeros(rows, 2, C, device="cuda")
    out, y, coef = co.conv3x3_bnin(x, sums, M, bn_b, 0.1, wt, stat_part=st)
    torch.cuda.synchronize()
    assert torch.equal(coef, coef_ref)
    assert torch.equal(bn_a.running_mean, bn_b.running_mean) and torch.equal(bn_a.running_var, bn_b.running_var)
    assert torch.equal(y, y_ref)
    assert torch.equal(out, out_ref)
    assert torch.allclose(st.sum(0), st_ref.sum(0), rtol=1e-4, atol=1e-2)
    # y_act not wanted: same output, nothing else written
    st2 = torch.zeros(rows, 2, C, device="cuda")
    out2, y2, _ = co.conv3x3_bnin(x, sums, M, copy.deepcopy(bn), 0.1, wt, stat_part=st2, want_act=False)
    assert y2 is None and torch.equal(out2, out_ref)


def test_bnin_against_torch_cpu():
    co, x, w, bn, sums = _case(128, 16, 256, 1)
    wt = co.weight_layout(w, 0)
    M = 128 * 256
    st = torch.zeros(co.stat_tiles(M), 2, 256, device="cuda")
    ref_bn = torch.nn.BatchNorm2d(256)
    ref_bn.load_state_dict({k: v.cpu() for k, v in bn.state_dict().items()})
    ref_bn.train()
    out, y, coef = co.conv3x3_bnin(x, sums, M, bn, 0.1, wt, stat_part=st)
    xc = x.float().cpu().permute(0, 3, 1, 2)
    with torch.no_grad():
        yr = torch.relu(ref_bn(xc))
        sel = [0, 1, 63, 127]                                     # the convolution on four images (the BatchNorm saw all 128)
        outr = torch.nn.functional.conv2d(yr[sel].to(BF16).float(), w.cpu().to(BF16).float(), padding=1)
    assert torch.allclose(bn.running_mean.cpu(), ref_bn.running_mean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(bn.running_var.cpu(), ref_bn.running_var, rtol=1e-5, atol=1e-5)
    yg = y.float().cpu().permute(0, 3, 1, 2)
    assert (yg - yr).abs().max() <= 2 ** -8 * yr.abs().max() + 1e-3           # one bf16 ulp of the stored activation
    og = out[sel].float().cpu().permute(0, 3, 1, 2)
    assert (og - outr).abs().max() <= 2 ** -7 * outr.abs().max() + 2e-2
    # the statistics the NEXT BatchNorm folds: sums of the bf16 output
    tot = st.sum(0).cpu()
    of = out.float().view(-1, 256)
    assert torch.allclose(tot[0], of.sum(0).cpu(), rtol=2e-3, atol=2.0)
    assert torch.allclose(tot[1], (of ** 2).sum(0).cpu(), rtol=2e-3, atol=2.0)
