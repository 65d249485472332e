"""GPU parity of the MFMA implicit-GEMM convolution (forward + data gradient) against torch CPU fp32 conv2d on the
same bf16-rounded operands.  Tolerances: f32-accumulator output 1e-3 rel (north_star); bf16 output adds one bf16
rounding (2^-8)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

CASES = [  # (B, H, Cin, Cout, k, stride)
    (2, 16, 64, 64, 1, 1), (2, 16, 64, 256, 1, 1), (3, 16, 128, 128, 3, 1), (2, 16, 128, 128, 3, 2),
    (2, 16, 256, 512, 1, 2), (2, 8, 512, 128, 1, 1), (5, 7, 64, 192, 3, 1),
]


def _mk(B, H, Cin, Cout, k, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, H, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).bfloat16().float()
    return x, w


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride", CASES)
def test_conv_forward_and_stats(B, H, Cin, Cout, k, stride):
    import ppv_amd.convops as co
    x, w = _mk(B, H, Cin, Cout, k)
    pad = (k - 1) // 2
    want = F.conv2d(x, w, stride=stride, padding=pad).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()
    wt = co.weight_layout(w.cuda(), 0)
    got32 = co.conv_fwd(xd, wt, stride, pad, out_f32=True)
    assert rel_err(got32, want) < 1e-3
    part = torch.zeros((co.stat_tiles(want.numel() // Cout), 2, Cout), device="cuda")
    got = co.conv_fwd(xd, wt, stride, pad, stat_part=part)
    assert rel_err(got.float(), want) < 2 ** -8 + 1e-3
    gf = got.float().reshape(-1, Cout)
    assert rel_err(part.sum(0)[0], gf.sum(0)) < 1e-4 and rel_err(part.sum(0)[1], (gf * gf).sum(0)) < 1e-4


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride", CASES)
def test_conv_dgrad(B, H, Cin, Cout, k, stride):
    import ppv_amd.convops as co
    x, w = _mk(B, H, Cin, Cout, k)
    pad = (k - 1) // 2
    x.requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).bfloat16().float()
    y.backward(g)
    want = x.grad.permute(0, 2, 3, 1)
    gd = g.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()
    wd = co.weight_layout(w.cuda(), 1)
    got = co.conv_dgrad(gd, wd, stride, pad, (H, H), out_f32=True)
    assert rel_err(got, want) < 1e-3
    add = torch.randn(want.shape, generator=torch.Generator().manual_seed(9)).bfloat16()
    got2 = co.conv_dgrad(gd, wd, stride, pad, (H, H), addend=add.cuda())
    assert rel_err(got2.float(), want + add.float()) < 2 ** -8 + 1e-3
    # ReLU mask of the conv input folded into the store (1 bit per element): lanes whose bit is clear are exactly zero
    act = torch.relu(torch.randn(want.shape, generator=torch.Generator().manual_seed(11))).bfloat16()
    act.view(-1)[::7] = -0.0
    keep = (act.float() > 0).cuda()
    bits = (keep.reshape(-1, 8).to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)).sum(-1).to(torch.uint8)
    got3 = co.conv_dgrad(gd, wd, stride, pad, (H, H), addend=add.cuda(), relu_bits=bits)
    assert torch.equal(got3, torch.where(keep, got2, torch.zeros_like(got2)))
    got4 = co.conv_dgrad(gd, wd, stride, pad, (H, H), relu_bits=bits)
    assert rel_err(got4.float(), want * keep.cpu()) < 2 ** -8 + 1e-3
