"""BASELINE.json full sizes (B = 128 trunk shapes, 512^2 FD camera at B = 32, 64x64x256 RAFT maps, 128 x 36x36x2048 decoder input)
checked through size-independent properties -- the oracle cannot run these in seconds, the identities can:
  * forward / data-gradient / weight-gradient of a convolution are mutually adjoint:  <conv(x, w), g> = <x, dgrad(g, w)> = <w, wgrad(g, x)>;
  * a circular convolution with a PSF preserves the image sum times the PSF sum (FD camera, before its per-image normalisation);
  * the two independent correlation implementations (materialised volume + pyramid, on-the-fly windows) agree;
  * attention weights sum to one per step and finished captions produce exact zeros."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,cout,k,stride,h", [(256, 1024, 1, 1, 16), (256, 256, 3, 1, 16), (512, 512, 3, 2, 16), (128, 128, 3, 1, 32)])
def test_conv_adjoint_identities_at_batch_128(cin, cout, k, stride, h):
    import ppv_amd.convops as co
    B, pad = 128, (k - 1) // 2
    g0 = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, h, h, cin, device="cuda", generator=g0).bfloat16()
    w = (torch.randn(cout, cin, k, k, device="cuda", generator=g0) / (cin * k * k) ** 0.5).bfloat16().float()
    ho = (h + 2 * pad - k) // stride + 1
    g = torch.randn(B, ho, ho, cout, device="cuda", generator=g0).bfloat16()
    y = co.conv_fwd(x, co.weight_layout(w, 0), stride, pad, out_f32=True)                     # f32 accumulators, no output rounding
    dx = co.conv_dgrad(g, co.weight_layout(w, 1), stride, pad, (h, h), out_f32=True)
    dw = co.conv_wgrad(g, x, k, k, stride, pad)
    a = (y.double() * g.double()).sum()
    b = (dx.double() * x.double()).sum()
    c = (dw.double() * w.double()).sum()
    scale = (y.double().norm() * g.double().norm()).item()
    assert abs((a - b).item()) < 1e-5 * scale and abs((a - c).item()) < 1e-5 * scale, (a.item(), b.item(), c.item())


def test_fd_circular_conv_conserves_mass_512_b32():
    import ppv_amd.fftconv as fc
    B, N = 32, 512
    g0 = torch.Generator(device="cuda").manual_seed(1)
    img = torch.rand(B, 3, N, N, device="cuda", generator=g0)
    psf = torch.rand(3, N, N, device="cuda", generator=g0)
    psf = psf / psf.sum((1, 2), keepdim=True) * torch.tensor([0.56, 0.28, 0.16], device="cuda").view(3, 1, 1)
    out, _, _ = fc.fftconv_fwd(img, fc.otf_build(psf, N, N), mode=1)          # circular conv, before the per-image normalisation
    want = img.double().sum((2, 3)) * psf.double().sum((1, 2))[None]
    assert ((out.double().sum((2, 3)) - want).abs() / want).max().item() < 1e-5


def test_two_correlation_implementations_agree_at_config4_size():
    from ppv_amd.raft_corr import CorrBlock, AlternateCorrBlock
    g0 = torch.Generator(device="cuda").manual_seed(2)
    f1 = torch.randn(1, 256, 64, 64, device="cuda", generator=g0)
    f2 = torch.randn(1, 256, 64, 64, device="cuda", generator=g0)
    ys, xs = torch.meshgrid(torch.arange(64, device="cuda"), torch.arange(64, device="cuda"), indexing="ij")
    coords = torch.stack([xs, ys], 0).float()[None] + 5.0 * torch.randn(1, 2, 64, 64, device="cuda", generator=g0)
    a = CorrBlock(f1, f2, num_levels=4, radius=4)(coords)
    b = AlternateCorrBlock(f1, f2, num_levels=4, radius=4)(coords)
    assert a.shape == b.shape == (1, 324, 64, 64)
    assert ((a - b).abs().max() / a.abs().max()).item() < 1e-4


def test_decoder_full_size_invariants():
    import ppv_amd.decoder as pd
    torch.manual_seed(0)
    B, V = 128, 9490
    dec = pd.DecoderWithAttention(512, 512, 512, V, encoder_dim=2048, dropout=0.5).cuda().eval()
    enc = torch.randn(B, 36, 36, 2048, device="cuda")
    caps = torch.randint(0, V, (B, 52), device="cuda")
    caplens = torch.randint(9, 19, (B, 1), device="cuda")
    with torch.no_grad():
        preds, caps_sorted, dec_len, alphas, order = dec(enc, caps, caplens)
    assert preds.shape == (B, max(dec_len), V) and alphas.shape == (B, max(dec_len), 1296)
    assert dec_len == sorted(dec_len, reverse=True) and sorted(order.tolist()) == list(range(B))
    live = torch.tensor([[1.0 if t < l else 0.0 for t in range(max(dec_len))] for l in dec_len], device="cuda")
    assert ((alphas.sum(-1) - live).abs().max()).item() < 1e-4                  # softmax rows sum to 1, finished captions are 0
    assert not preds[live == 0].any() and torch.isfinite(preds).all()
