"""GPU parity of the trunk kernels around the convolutions: weight gradient, stem conv (+ data gradient), train-mode
BatchNorm (+ residual + ReLU) forward/backward, stem max-pool and the adaptive average pool -- each against torch
CPU fp32 on the same bf16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = 2 ** -8 + 1e-3      # one bf16 rounding of the stored result + north_star 1e-3


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def r16(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride", [(4, 16, 128, 128, 3, 1), (2, 16, 256, 128, 1, 1), (3, 16, 128, 256, 3, 2),
                                                   (2, 8, 256, 512, 1, 2), (9, 8, 128, 128, 3, 1), (2, 32, 128, 256, 3, 1),
                                                   (3, 16, 256, 256, 3, 1), (1, 64, 128, 128, 3, 1)])
def test_conv_wgrad(B, H, Cin, Cout, k, stride):
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(0)
    x = r16(torch.randn(B, Cin, H, H, generator=g0))
    w = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    pad = (k - 1) // 2
    y = F.conv2d(x, w, stride=stride, padding=pad)
    g = r16(torch.randn(y.shape, generator=g0))
    y.backward(g)
    got = co.conv_wgrad(nhwc(g).cuda().bfloat16(), nhwc(x).cuda().bfloat16(), k, k, stride, pad)
    assert rel_err(got, w.grad) < 1e-3


def test_stem_forward_and_data_gradient():
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(0)
    B, H = 3, 64
    img = torch.rand(B, 3, H, H, generator=g0)
    w = r16(torch.randn(64, 3, 7, 7, generator=g0) * 0.1)
    x = r16(img).requires_grad_(True)                       # the kernel rounds the f32 sensor image to bf16 operands
    y = F.conv2d(x, w, stride=2, padding=3)
    tiles = co.stat_tiles(B * (H // 2) ** 2)
    part = torch.zeros(tiles, 2, 64, device="cuda")
    got = co.stem_conv(img.cuda(), co.stem_weight_layout(w.cuda(), 0), part)
    assert rel_err(got.float(), nhwc(y)) < BF
    gf = got.float().reshape(-1, 64)
    assert rel_err(part.sum(0)[0], gf.sum(0)) < 1e-4 and rel_err(part.sum(0)[1], (gf * gf).sum(0)) < 1e-4
    g = r16(torch.randn(y.shape, generator=g0))
    y.backward(g)
    gd = co.stem_dgrad(nhwc(g).cuda().bfloat16(), co.stem_weight_layout(w.cuda(), 1))
    assert rel_err(gd, x.grad) < 1e-3


@pytest.mark.parametrize("B,H,W", [(2, 64, 256), (1, 32, 512)])
def test_stem_forward_row_staged(B, H, W):
    """W/2 % 128 == 0 takes the row-staged kernel (one output-row segment per tile): same result as the reference conv on the
    bf16-rounded operands, statistics included; image borders and the segment seam (W = 512) are inside the case."""
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(1)
    img = torch.rand(B, 3, H, W, generator=g0)
    w = r16(torch.randn(64, 3, 7, 7, generator=g0) * 0.1)
    y = F.conv2d(r16(img), w, stride=2, padding=3)
    part = torch.zeros(co.stat_tiles(B * (H // 2) * (W // 2)), 2, 64, device="cuda")
    got = co.stem_conv(img.cuda(), co.stem_weight_layout(w.cuda(), 0), part)
    assert rel_err(got.float(), nhwc(y)) < BF
    assert (got.float().cpu() - nhwc(y)).abs().max().item() < 0.05
    gf = got.float().reshape(-1, 64)
    assert rel_err(part.sum(0)[0], gf.sum(0)) < 1e-4 and rel_err(part.sum(0)[1], (gf * gf).sum(0)) < 1e-4
    # data gradient: the row-staged single launch against autograd of the same conv
    x = r16(img).requires_grad_(True)
    y2 = F.conv2d(x, w, stride=2, padding=3)
    g = r16(torch.randn(y2.shape, generator=g0))
    y2.backward(g)
    gd = co.stem_dgrad(nhwc(g).cuda().bfloat16(), co.stem_weight_layout(w.cuda(), 1))
    assert rel_err(gd, x.grad) < 1e-3
    assert (gd.cpu() - x.grad).abs().max().item() < 1e-2 * x.grad.abs().max().item()


@pytest.mark.parametrize("C,res_mode", [(64, 0), (256, 1), (512, 2), (2048, 1)])
def test_batchnorm_train_forward_backward(C, res_mode):
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(1)
    B, H = 4, 8
    x = r16(torch.randn(B, C, H, H, generator=g0) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C, generator=g0) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g0) * 0.1).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    res = r16(torch.randn(B, C, H, H, generator=g0)) if res_mode else None
    z = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if res_mode == 1:
        z = z + res
    if res_mode == 2:
        g2, b2 = torch.rand(C, generator=g0) + 0.5, torch.randn(C, generator=g0) * 0.1
        z = z + res * g2.view(1, C, 1, 1) + b2.view(1, C, 1, 1)
    y = F.relu(z)
    gy = r16(torch.randn(y.shape, generator=g0))
    y.backward(gy)
    # product: statistics come from the conv epilogue; emulate its partials from the bf16 tensor
    xd = nhwc(x.detach()).cuda().bfloat16()
    xf = xd.float().reshape(-1, C)
    part = torch.stack([xf.sum(0), (xf * xf).sum(0)])[None].contiguous()
    rmd, rvd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    coef = co.bn_finalize(part, xf.shape[0], gamma.detach().cuda(), beta.detach().cuda(), rmd, rvd)
    assert rel_err(rmd, rm) < 1e-4 and rel_err(rvd, rv) < 1e-4
    resd = nhwc(res).cuda().bfloat16() if res_mode else None
    coef2 = torch.stack([g2, b2, g2, g2]).cuda().contiguous() if res_mode == 2 else None
    yd, bits = co.bn_act(xd, coef, resd, coef2, relu=True, want_bits=True)
    assert rel_err(yd.float(), nhwc(y.detach())) < BF
    want_bits = ((yd.reshape(-1, 8) > 0).to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(bits, want_bits)                       # bit k of byte i <-> (y[8 i + k] > 0), exactly
    gx, gpre, dg, db = co.bn_bwd(nhwc(gy).cuda().bfloat16(), yd, xd, coef, relu=True, want_gpre=True)
    assert rel_err(gx.float(), nhwc(x.grad)) < 2 * BF
    assert rel_err(dg, gamma.grad) < 5e-3 and rel_err(db, beta.grad) < 5e-3
    mask = (nhwc(y.detach()) > 0).float()
    assert rel_err(gpre.float() * mask.cuda(), nhwc(gy) * mask) < BF
    if res_mode == 0:          # mask recomputed from x and the BN coefficients must give the same result
        gx2, _, dg2, db2 = co.bn_bwd(nhwc(gy).cuda().bfloat16(), None, xd, coef, relu=2)
        assert torch.equal(gx2, gx) and torch.equal(dg2, dg)


def test_stem_bn_relu_maxpool_forward_backward():
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(2)
    B, H, C = 2, 16, 64
    x = r16(torch.randn(B, C, H, H, generator=g0))
    scale, shift = torch.rand(C, generator=g0) + 0.5, torch.randn(C, generator=g0) * 0.2
    a = r16(F.relu(x * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1))).requires_grad_(True)
    y = F.max_pool2d(a, 3, stride=2, padding=1)
    gy = r16(torch.randn(y.shape, generator=g0))
    y.backward(gy)
    coef = torch.stack([scale, shift, scale, scale]).cuda().contiguous()
    yd, arg = co.bn_relu_maxpool(nhwc(x).cuda().bfloat16(), coef)
    assert rel_err(yd.float(), nhwc(y.detach())) < 1e-6
    gpre = co.maxpool_relu_bwd(nhwc(gy).cuda().bfloat16(), yd, arg, (H, H))
    want = nhwc(a.grad * (a.detach() > 0))
    assert rel_err(gpre.float(), want) < BF


def test_adaptive_pool_8_to_36():
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(3)
    x = r16(torch.randn(2, 2048, 8, 8, generator=g0)).requires_grad_(True)
    y = F.adaptive_avg_pool2d(x, 36)
    gy = torch.randn(y.shape, generator=g0)
    y.backward(gy)
    yd = co.adaptive_pool_fwd(nhwc(x.detach()).cuda().bfloat16(), 36)
    assert yd.shape == (2, 36, 36, 2048) and yd.dtype == torch.float32
    assert rel_err(yd, nhwc(y.detach())) < 1e-6
    gx = co.adaptive_pool_bwd(nhwc(gy).cuda(), (8, 8))
    assert rel_err(gx.float(), nhwc(x.grad)) < BF


@pytest.mark.parametrize("C,rows", [(256, 4096), (512, 1000), (2048, 130)])
def test_bn_backward_takes_a_second_batchnorms_sums(C, rows):
    """bn_bwd(..., sums2=(x2, part2)): same g_x / dgamma / dbeta as the plain call, and part2 holds sum g and sum g * x2 so that
    the second BatchNorm's backward (part_ready) equals its own full backward."""
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(C)
    g = torch.randn(rows, C, generator=g0).bfloat16().cuda()
    x = torch.randn(rows, C, generator=g0).bfloat16().cuda()
    x2 = torch.randn(rows, C, generator=g0).bfloat16().cuda()
    mk = lambda: torch.stack([torch.rand(C, generator=g0) + 0.5, torch.randn(C, generator=g0) * 0.1,
                              torch.randn(C, generator=g0) * 0.1, torch.rand(C, generator=g0) + 0.5]).cuda().contiguous()
    coef, coef2 = mk(), mk()
    want = co.bn_bwd(g, None, x, coef, 0)
    part2 = torch.zeros(64 * C, device="cuda")
    got = co.bn_bwd(g, None, x, coef, 0, sums2=(x2, part2))
    # two runs of the reduce kernel add their f32 atomics in different orders: equal up to that, not bit for bit
    assert rel_err(got[0].float(), want[0].float()) < 2 ** -9 and rel_err(got[2], want[2]) < 1e-5 and rel_err(got[3], want[3]) < 1e-5
    sums = part2[:16 * C].view(8, 2, C).sum(0)
    assert rel_err(sums[0], g.float().sum(0)) < 1e-4 and rel_err(sums[1], (g.float() * x2.float()).sum(0)) < 1e-4
    want2 = co.bn_bwd(g, None, x2, coef2, 0)
    got2 = co.bn_bwd(g, None, x2, coef2, 0, part=part2, part_ready=True)
    assert rel_err(got2[0].float(), want2[0].float()) < 2 ** -7 and rel_err(got2[2], want2[2]) < 1e-4 and rel_err(got2[3], want2[3]) < 1e-4


@pytest.mark.parametrize("m,N,K,strided", [(128, 2560, 512, True), (37, 2048, 3072, False), (300, 9490, 512, False), (5, 48, 336, True)])
def test_linear_f32_is_exact_f32(m, N, K, strided):
    """csrc/gemm_f32.hip (decoder dense layers, models.py:199-214): v_mfma_f32_16x16x4_f32 is an f32 fmaf chain -- the result
    must agree with a float64 reference to f32 rounding of the accumulation (1e-6 of the row's absolute mass), far below bf16."""
    import ppv_amd.convops as co
    g = torch.Generator().manual_seed(0)
    xw = torch.randn(m, K + 64, generator=g).cuda()
    x = xw[:, 32:32 + K] if strided else xw[:, :K].contiguous()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    out = torch.full((m, N + 8), 7.0, device="cuda")[:, :N] if strided else None
    got = co.linear_f32(x, w, b, out=out)
    want = x.double() @ w.double().t() + b.double()
    mass = (x.double().abs() @ w.double().abs().t() + b.double().abs())
    assert ((got.double() - want).abs() / mass).max().item() < 2e-6
    if strided:
        assert got.data_ptr() == out.data_ptr()


@pytest.mark.parametrize("rows,C,T", [(4096, 64, 32), (2048, 256, 16), (1031, 1024, 8), (520, 2048, 4), (37, 128, 1), (32768, 256, 32)])
@pytest.mark.parametrize("res_mode", [0, 1, 2])
def test_batchnorm_train_single_launch(rows, C, T, res_mode):
    """co.bn_act_train (statistics -> coefficients -> apply in one launch) against torch BatchNorm2d(training) on the CPU: the stored
    tensor within one bf16 rounding, coefficients and running statistics at f32 accuracy, ReLU bits consistent with the stored tensor."""
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(C + rows)
    x = r16(torch.randn(rows, C, generator=g0) * 1.5 + 0.3)
    res = r16(torch.randn(rows, C, generator=g0))

    def mk_bn():
        bn = torch.nn.BatchNorm2d(C)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, generator=g0) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=g0) * 0.2)
            bn.running_mean.copy_(torch.randn(C, generator=g0) * 0.1)
            bn.running_var.copy_(torch.rand(C, generator=g0) + 0.5)
        return bn.train()

    def part_of(t):                                          # [T][2][C] partial sums as the conv epilogue leaves them (any split)
        idx = torch.arange(rows) % T
        p = torch.zeros(T, 2, C)
        p[:, 0].index_add_(0, idx, t)
        p[:, 1].index_add_(0, idx, t * t)
        return p

    bn1, bn2 = mk_bn(), mk_bn()
    ref1, ref2 = mk_bn(), mk_bn()
    for a, b in ((bn1, ref1), (bn2, ref2)):
        b.load_state_dict(a.state_dict())
    as4 = lambda t: t.t().reshape(1, C, rows, 1)
    want = ref1(as4(x))
    if res_mode == 1:
        want = want + as4(res)
    if res_mode == 2:
        want = want + ref2(as4(res))
    want = torch.relu(want).reshape(C, rows).t()
    d1, d2 = bn1.cuda(), bn2.cuda()
    y, bits, coef, coef2 = co.bn_act_train(x.cuda().bfloat16(), part_of(x).cuda(), rows, d1, 0.1, res=None if res_mode == 0 else res.cuda().bfloat16(),
                                            res_stats=(part_of(res).cuda(), d2, 0.1) if res_mode == 2 else None, want_bits=True)
    assert rel_err(y.float(), want) < BF
    assert rel_err(d1.running_mean.cpu(), ref1.running_mean) < 1e-5 and rel_err(d1.running_var.cpu(), ref1.running_var) < 1e-5
    mean, var = x.mean(0), x.var(0, unbiased=False)
    assert rel_err(coef[2].cpu(), mean) < 1e-5 and rel_err(coef[3].cpu(), torch.rsqrt(var + 1e-5)) < 1e-5
    assert rel_err(coef[0].cpu(), ref1.weight.detach() * torch.rsqrt(var + 1e-5)) < 1e-5
    if res_mode == 2:
        assert rel_err(d2.running_var.cpu(), ref2.running_var) < 1e-5 and rel_err(coef2[2].cpu(), res.mean(0)) < 1e-5
    k = (bits.view(-1, 1).int() >> torch.arange(8, device="cuda").int()) & 1
    assert torch.equal(k.view(-1).bool(), (y > 0).view(-1))


@pytest.mark.parametrize("rows,C,T", [(4096, 256, 1), (4096, 256, 2), (1000, 64, 2), (520, 128, 3), (2048, 512, 2), (1028, 1024, 2),
                                      (1028, 1024, 4), (517, 2048, 2), (1024, 256, 6)])
def test_bn_act_fold_equals_finalize_plus_act(rows, C, T):
    """ppv_bn_act_fold_rows (the step's default since round 3: statistics in T partial rows, every workgroup derives the coefficients
    itself, no ppv_bn_finalize launch) against ppv_bn_finalize + ppv_bn_act on the same sums: same activations (one bf16 ulp),
    coefficients and running statistics; ragged row counts (last workgroup partly empty), every channel width of the trunk."""
    import ppv_amd.convops as co
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).bfloat16().cuda().view(1, rows, 1, C)
    res = torch.randn(rows, C, generator=g).bfloat16().cuda().view(1, rows, 1, C)
    xf = x.float().view(rows, C)
    # T partial rows that add up to the column sums (the convolution's row tiles fold into row tile % T)
    parts = [xf[t::T] for t in range(T)]
    sums = torch.stack([torch.stack([p_.sum(0), (p_ * p_).sum(0)]) for p_ in parts]).contiguous()
    bn_a, bn_b = torch.nn.BatchNorm2d(C).cuda(), torch.nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        for bn in (bn_a, bn_b):
            bn.weight.copy_(torch.rand(C, generator=torch.Generator().manual_seed(1)) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.2)
    coef = co.bn_finalize(sums, rows, bn_a.weight.detach(), bn_a.bias.detach(), bn_a.running_mean, bn_a.running_var, 0.1, bn_a.eps)
    want, wbits = co.bn_act(x, coef, res=res, want_bits=True)
    got, gbits, gcoef = co.bn_act_fold(x, sums, rows, bn_b, 0.1, res=res, want_bits=True)
    assert rel_err(gcoef, coef) < 1e-6
    assert rel_err(got.float(), want.float()) < 2 ** -7 and (gbits != wbits).float().mean().item() < 1e-3
    assert rel_err(bn_b.running_mean, bn_a.running_mean) < 1e-6 and rel_err(bn_b.running_var, bn_a.running_var) < 1e-6
    got0, _, _ = co.bn_act_fold(x, sums, rows, bn_b, 0.1, relu=False)
    assert rel_err(got0.float(), co.bn_act(x, coef, relu=False).float()) < 2 ** -7


@pytest.mark.parametrize("B,H,C", [(2, 16, 64), (3, 24, 64), (1, 8, 128)])
def test_fused_stem_backward_equals_autograd(B, H, C):
    """ppv_maxpool_bn_bwd (round 3): pooled gradient -> gradient of the raw stem output, against torch autograd of
    MaxPool2d(3,2,1) o ReLU o BatchNorm2d(train) (models.py:17-21 / train.py:245) and against the two-kernel path it replaces."""
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(4)
    x = r16(torch.randn(B, C, H, H, generator=g0) * 1.5 + 0.3).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g0) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g0) * 0.2)
    xd = nhwc(x.detach()).cuda().bfloat16()
    xf = xd.float().view(-1, C)
    sums = torch.stack([xf.sum(0), (xf * xf).sum(0)]).view(1, 2, C).contiguous()
    coef = co.bn_finalize(sums, xf.shape[0], bn.weight.detach().cuda(), bn.bias.detach().cuda(), None, None, 0.1, bn.eps)
    yd, arg = co.bn_relu_maxpool(xd, coef)
    gy = r16(torch.randn(B, C, H // 2, H // 2, generator=g0))
    gyd = nhwc(gy).cuda().bfloat16()
    gx, dg, db = co.maxpool_bn_bwd(gyd, yd, arg, xd, coef, want_affine=True)
    # the path it replaces (same kernels' arithmetic, pre-pool tensor rounded to bf16 in between)
    gpre = co.maxpool_relu_bwd(gyd, yd, arg, (H, H))
    gx2, _, dg2, db2 = co.bn_bwd(gpre, None, xd, coef, False, want_affine=True)
    assert rel_err(gx.float(), gx2.float()) < 2 * BF and rel_err(dg, dg2) < 5e-3 and rel_err(db, db2) < 5e-3   # one bf16 rounding fewer (the sums see the unrounded gather)
    # torch autograd on the same bf16-rounded activation chain
    a = r16(F.relu(bn(x)))
    a.retain_grad()
    F.max_pool2d(a, 3, stride=2, padding=1).backward(gy)
    assert rel_err(gx.float(), nhwc(x.grad)) < 3e-2          # the product's ReLU / arg-max decisions were taken on bf16 values
    assert rel_err(db, bn.bias.grad) < 3e-2 and rel_err(dg, bn.weight.grad) < 3e-2


def test_fused_stem_backward_with_near_dead_channels():
    """r3 advisor: the stem backward rebuilds the raw value from the pooled bf16 activation, x = (y - shift) / scale.  ImageNet-pretrained
    bn1 (what models.py:17 loads) has gammas down to ~1e-8 beside betas of order 0.1: there the reconstruction is noise.  Such channels
    must take x from the raw tensor (arg-max gather): gradients against torch autograd at the usual tolerance for EVERY gamma."""
    import ppv_amd.convops as co
    B, H, C = 2, 32, 64
    g0 = torch.Generator().manual_seed(7)
    x = r16(torch.randn(B, C, H, H, generator=g0) * 1.5 + 0.3).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C).train()
    with torch.no_grad():
        gam = torch.rand(C, generator=g0) + 0.5
        gam[0::4] = 1e-8
        gam[1::4] = 1e-3
        gam[2::4] = 0.05
        bn.weight.copy_(gam)
        bn.bias.copy_(torch.rand(C, generator=g0) * 0.3 + 0.05)            # positive: the ReLU passes the near-dead channels
    xd = nhwc(x.detach()).cuda().bfloat16()
    xf = xd.float().view(-1, C)
    sums = torch.stack([xf.sum(0), (xf * xf).sum(0)]).view(1, 2, C).contiguous()
    coef = co.bn_finalize(sums, xf.shape[0], bn.weight.detach().cuda(), bn.bias.detach().cuda(), None, None, 0.1, bn.eps)
    yd, arg = co.bn_relu_maxpool(xd, coef)
    gy = r16(torch.randn(B, C, H // 2, H // 2, generator=g0))
    gyd = nhwc(gy).cuda().bfloat16()
    gx, dg, db = co.maxpool_bn_bwd(gyd, yd, arg, xd, coef, want_affine=True)
    # the two-kernel path reads the raw tensor everywhere: the reference for what the sums must be
    gpre = co.maxpool_relu_bwd(gyd, yd, arg, (H, H))
    gx2, _, dg2, db2 = co.bn_bwd(gpre, None, xd, coef, False, want_affine=True)
    for sel in (slice(0, None, 4), slice(1, None, 4), slice(2, None, 4), slice(3, None, 4)):
        assert rel_err(dg[sel], dg2[sel]) < 2e-2, sel
        assert rel_err(db[sel], db2[sel]) < 5e-3, sel
        assert rel_err(gx.float()[..., sel], gx2.float()[..., sel]) < 3e-2, sel
