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


RED_CASES = [  # (B, H, Cout_of_conv = K of the data gradient, Cin = N columns): the three production tiles + a ragged M
    (2, 16, 64, 256), (8, 32, 128, 512), (64, 32, 64, 256), (32, 32, 64, 256), (16, 16, 256, 1024), (3, 7, 128, 256),
]


@pytest.mark.parametrize("B,H,Cout,Cin", RED_CASES)
@pytest.mark.parametrize("with_addend", [True, False])
def test_dgrad_takes_bn_backward_sums(B, H, Cout, Cin, with_addend):
    """ppv_conv_gemm_red: the stored tensor is bit-identical to the plain launch; its per-column sums (sum g, sum g * x)
    equal those of the stored tensor; bn_bwd(part_ready) equals bn_bwd on the same gradient."""
    import ppv_amd.convops as co
    gen = torch.Generator().manual_seed(3)
    g = torch.randn(B, H, H, Cout, generator=gen).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=gen) / Cout ** 0.5).cuda()
    wd = co.weight_layout(w, 1)
    add = torch.randn(B, H, H, Cin, generator=gen).bfloat16().cuda() if with_addend else None
    act = torch.relu(torch.randn(B, H, H, Cin, generator=gen))
    keep = (act > 0).cuda()
    bits = (keep.reshape(-1, 8).to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)).sum(-1).to(torch.uint8)
    xraw = torch.randn(B, H, H, Cin, generator=gen).bfloat16().cuda()
    plain = co.conv_dgrad(g, wd, 1, 0, (H, H), addend=add, relu_bits=bits)
    part = torch.zeros(64 * Cin, device="cuda")
    fused = co.conv_dgrad(g, wd, 1, 0, (H, H), addend=add, relu_bits=bits, red=(xraw, part))
    assert torch.equal(plain, fused)
    sums = part[:co.RED_ROWS * 2 * Cin].view(co.RED_ROWS, 2, Cin).sum(0)
    gf, xf = plain.float().reshape(-1, Cin), xraw.float().reshape(-1, Cin)
    assert rel_err(sums[0], gf.sum(0)) < 1e-4
    assert rel_err(sums[1], (gf * xf).sum(0)) < 1e-4
    assert part[co.RED_ROWS * 2 * Cin:].abs().max().item() == 0.0
    coef = torch.stack([torch.rand(Cin, generator=gen) + 0.5, torch.randn(Cin, generator=gen) * 0.1,
                        torch.randn(Cin, generator=gen) * 0.1, torch.rand(Cin, generator=gen) + 0.5]).cuda().contiguous()
    want = co.bn_bwd(plain, None, xraw, coef, 0)
    got = co.bn_bwd(fused, None, xraw, coef, 0, part=part, part_ready=True)
    assert rel_err(got[0].float(), want[0].float()) < 2 ** -7
    assert rel_err(got[2], want[2]) < 1e-4 and rel_err(got[3], want[3]) < 1e-4


@pytest.mark.parametrize("B,H,Cout,Cin,k,stride", [(2, 16, 256, 128, 1, 1), (32, 32, 256, 256, 1, 1), (64, 32, 128, 128, 3, 1),
                                                    (4, 16, 128, 256, 3, 2), (3, 7, 512, 128, 1, 1), (32, 64, 256, 64, 1, 1),
                                                    (33, 64, 64, 64, 3, 1)])
def test_dgrad_relu_recompute_and_sums(B, H, Cout, Cin, k, stride):
    """red=(x, part, coef): the stored gradient is the plain one with the BN + ReLU mask (x * scale + shift > 0) applied, the
    sums are those of the stored tensor, and bn_bwd(relu=0, part_ready) equals bn_bwd(relu=2) on the unmasked gradient."""
    import ppv_amd.convops as co
    gen = torch.Generator().manual_seed(5)
    Ho = (H + 2 * ((k - 1) // 2) - k) // stride + 1
    g = torch.randn(B, Ho, Ho, Cout, generator=gen).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, k, k, generator=gen) / (Cout * k * k) ** 0.5).cuda()
    wd = co.weight_layout(w, 1)
    xraw = torch.randn(B, H, H, Cin, generator=gen).bfloat16().cuda()
    coef = torch.stack([torch.rand(Cin, generator=gen) + 0.5, torch.randn(Cin, generator=gen) * 0.3,
                        torch.randn(Cin, generator=gen) * 0.1, torch.rand(Cin, generator=gen) + 0.5]).cuda().contiguous()
    pad = (k - 1) // 2
    plain = co.conv_dgrad(g, wd, stride, pad, (H, H))
    part = torch.zeros(64 * Cin, device="cuda")
    fused = co.conv_dgrad(g, wd, stride, pad, (H, H), red=(xraw, part, coef))
    keep = torch.addcmul(coef[1], xraw.float(), coef[0]) > 0          # fma, as the kernels compute it
    assert torch.equal(fused, torch.where(keep, plain, torch.zeros_like(plain)))
    sums = part[:co.RED_ROWS * 2 * Cin].view(co.RED_ROWS, 2, Cin).sum(0)
    gf, xf = fused.float().reshape(-1, Cin), xraw.float().reshape(-1, Cin)
    assert rel_err(sums[0], gf.sum(0)) < 1e-4 and rel_err(sums[1], (gf * xf).sum(0)) < 1e-4
    want = co.bn_bwd(plain, None, xraw, coef, 2)
    got = co.bn_bwd(fused, None, xraw, coef, 0, part=part, part_ready=True)
    assert rel_err(got[0].float(), want[0].float()) < 2 ** -7
    assert rel_err(got[2], want[2]) < 1e-4 and rel_err(got[3], want[3]) < 1e-4


def test_cooperative_conv_bn_relu_equals_the_two_launch_form():
    """ppv_conv_bn_relu_coop (round-4 experiment, DESIGN 4d: measured 11.5 us slower than the two launches it fuses, not on the product
    path): conv 1x1 + train-mode BatchNorm + ReLU with a grid barrier inside the launch.  Must reproduce ppv_conv_gemm + ppv_bn_act_fold_rows:
    raw tensor bit for bit, activation / coefficients / running statistics to rounding; the barrier must complete (no timeout flag), and a
    grid that cannot be resident at once must be refused."""
    import ppv_amd.convops as co
    from ppv_amd import _lib
    from ppv_amd._lib import check, ptr, stream_ptr
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    B, H, CIN, COUT, ROWS = 64, 16, 512, 256, 2
    M = B * H * H
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, CIN, generator=g).bfloat16().to(dev)
    w = co.weight_layout((torch.randn(COUT, CIN, 1, 1, generator=g) * 0.05).to(dev), 0)
    gamma = (torch.rand(COUT, generator=g) + 0.5).to(dev)
    beta = (torch.randn(COUT, generator=g) * 0.2).to(dev)
    zp = co.zero_page(dev)

    def run(coop):
        rm, rv = torch.zeros(COUT, device=dev), torch.ones(COUT, device=dev)
        x1, y1 = torch.empty(B, H, H, COUT, dtype=torch.bfloat16, device=dev), torch.empty(B, H, H, COUT, dtype=torch.bfloat16, device=dev)
        stats, coef = torch.zeros(ROWS, 2, COUT, device=dev), torch.empty(4, COUT, device=dev)
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        if coop:
            check(L.ppv_conv_bn_relu_coop(ptr(x), ptr(w), ptr(x1), ptr(y1), ptr(stats), ptr(counter), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1e-5,
                                          ptr(coef), ptr(zp), B, H, H, CIN, COUT, ROWS, stream_ptr()), "coop")
        else:
            check(L.ppv_conv_gemm(ptr(x), ptr(w), ptr(x1), ptr(stats), None, None, ptr(zp), B, H, H, CIN, H, H, COUT, 1, 1, 1, 0, 1, 0, ROWS, stream_ptr()), "conv")
            check(L.ppv_bn_act_fold_rows(ptr(x1), ptr(stats), ROWS, float(M), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.1, 1e-5, ptr(coef), None, ptr(y1),
                                         None, M * COUT, COUT, 0, 1, stream_ptr()), "bn")
        torch.cuda.synchronize()
        return x1, y1, coef, rm, rv, counter

    a, b = run(False), run(True)
    assert int(b[5][1]) == 0 and int(b[5][0]) == (M // 256) * (COUT // 128)
    assert torch.equal(a[0], b[0])
    assert rel_err(b[1].float(), a[1].float()) < 2 ** -7
    for i in (2, 3, 4):
        assert rel_err(b[i], a[i]) < 1e-5
    # 1024 tiles of 256 x 128 cannot be resident on 256 CUs at one per CU: refused before any launch
    big = torch.empty(1, device=dev)
    assert L.ppv_conv_bn_relu_coop(ptr(x), ptr(w), ptr(big), ptr(big), ptr(big), ptr(big), ptr(gamma), ptr(beta), None, None, 0.1, 1e-5, ptr(big), ptr(zp),
                                   128, 32, 32, CIN, 256, ROWS, stream_ptr()) == _lib.PPV_ERR_BAD_SIZE
