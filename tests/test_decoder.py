"""Attention decoder (Image_Caption/models.py:57-218): oracle vs the reference's golden vectors on CPU;
HIP module vs golden / oracle on the GPU (forward, loss and every gradient)."""
import os
import numpy as np
import pytest
import torch

from conftest import rel_err
from fan_fill import fill_by_name

GOLD = os.path.join(os.path.dirname(__file__), "golden", "decoder.npz")


def _oracle_from_golden(g):
    from oracle.decoder import DecoderWithAttention
    B, S, E, A, M, D, V, L = [int(v) for v in g["dims"]]
    dec = DecoderWithAttention(attention_dim=A, embed_dim=M, decoder_dim=D, vocab_size=V, encoder_dim=E, dropout=0.3).eval()
    fill_by_name(dec)
    with torch.no_grad():
        dec.embedding.weight.copy_(torch.from_numpy(g["emb_weight"]))
    return dec


def test_oracle_matches_reference_golden():
    from oracle.decoder import caption_loss
    g = np.load(GOLD)
    dec = _oracle_from_golden(g)
    enc = torch.from_numpy(g["enc"]).requires_grad_(True)
    preds, caps_sorted, dec_len, alphas, order = dec(enc, torch.from_numpy(g["caps"]), torch.from_numpy(g["caplens"]))
    assert dec_len == g["dec_len"].tolist() and order.tolist() == g["order"].tolist()          # integer paths: bit-exact
    assert rel_err(preds, g["preds"]) < 1e-5 and rel_err(alphas, g["alphas"]) < 1e-5
    loss = caption_loss(preds, caps_sorted, dec_len, alphas)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    assert rel_err(enc.grad, g["g_enc"]) < 1e-4
    for n, p in dec.named_parameters():
        if n == "attention.full_att.bias":                      # softmax is shift-invariant: the true gradient is 0
            assert p.grad.abs().max() < 1e-6
            continue
        assert rel_err(p.grad, g["g_" + n]) < 1e-4, n


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def _l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


@pytest.mark.gpu
def test_hip_decoder_matches_reference_golden():
    """Forward, loss and every gradient against the reference's own outputs.  Tolerances: the two streamed tables
    (att1, encoder_out) are bf16 (BASELINE.json config 3); everything else is f32."""
    import ppv_amd.decoder as pd
    from oracle.decoder import caption_loss
    g = np.load(GOLD)
    B, S, E, A, M, D, V, L = [int(v) for v in g["dims"]]
    dec = pd.DecoderWithAttention(attention_dim=A, embed_dim=M, decoder_dim=D, vocab_size=V, encoder_dim=E, dropout=0.3).eval()
    fill_by_name(dec)
    with torch.no_grad():
        dec.embedding.weight.copy_(torch.from_numpy(g["emb_weight"]))
    dec = dec.cuda()
    enc = torch.from_numpy(g["enc"]).cuda().requires_grad_(True)
    preds, caps_sorted, dec_len, alphas, order = dec(enc, torch.from_numpy(g["caps"]).cuda(), torch.from_numpy(g["caplens"]).cuda())
    assert dec_len == g["dec_len"].tolist() and order.tolist() == g["order"].tolist()
    assert preds.shape == g["preds"].shape and alphas.shape == g["alphas"].shape
    assert _l2(preds.detach().cpu(), g["preds"]) < 1e-2 and _l2(alphas.detach().cpu(), g["alphas"]) < 1e-2
    # positions past a caption's end are exactly zero (models.py:194-195)
    pz = preds.detach().cpu().numpy()
    for b, l in enumerate(dec_len):
        assert not pz[b, l:].any() and not alphas.detach().cpu().numpy()[b, l:].any()
    loss = caption_loss(preds.cpu(), caps_sorted.cpu(), dec_len, alphas.cpu())
    assert abs(loss.item() - float(g["loss"])) < 2e-2
    loss.backward()
    assert _l2(enc.grad.cpu(), g["g_enc"]) < 3e-2 and _cos(enc.grad.cpu(), g["g_enc"]) > 0.999
    for n, p in dec.named_parameters():
        if n == "attention.full_att.bias":
            assert p.grad.abs().max() == 0
            continue
        assert p.grad is not None, n
        assert _l2(p.grad.cpu(), g["g_" + n]) < 3e-2 and _cos(p.grad.cpu(), g["g_" + n]) > 0.999, (n, _l2(p.grad.cpu(), g["g_" + n]))


@pytest.mark.gpu
def test_hip_decoder_vs_oracle_ragged_and_train_mode():
    """A second shape (P not a multiple of the 48-pixel slab, batch shrinking to 1, equal lengths) against the oracle, and
    train-mode dropout statistics."""
    import ppv_amd.decoder as pd
    from oracle.decoder import DecoderWithAttention as Ref, caption_loss
    torch.manual_seed(1)
    B, E, A, M, D, V, L = 7, 256, 128, 32, 48, 40, 9
    ref = Ref(A, M, D, V, encoder_dim=E, dropout=0.5).eval()
    fill_by_name(ref)
    dec = pd.DecoderWithAttention(A, M, D, V, encoder_dim=E, dropout=0.5).eval()
    dec.load_state_dict(ref.state_dict())
    dec = dec.cuda()
    enc = torch.randn(B, 7, 9, E)                                 # P = 63
    caps = torch.randint(0, V, (B, L))
    caplens = torch.tensor([[9], [3], [3], [5], [2], [9], [6]])
    e1 = enc.clone().requires_grad_(True)
    p1, c1, d1, a1, o1 = ref(e1, caps, caplens)
    caption_loss(p1, c1, d1, a1).backward()
    e2 = enc.cuda().requires_grad_(True)
    p2, c2, d2, a2, o2 = dec(e2, caps.cuda(), caplens.cuda())
    assert d1 == d2 and sorted(o1.tolist()) == sorted(o2.tolist())
    # ties in the length sort may be ordered differently by the CPU and GPU sorts: compare per original image
    inv1, inv2 = torch.argsort(o1), torch.argsort(o2.cpu())
    assert _l2(p2.detach().cpu()[inv2], p1.detach()[inv1]) < 1e-2 and _l2(a2.detach().cpu()[inv2], a1.detach()[inv1]) < 1e-2
    caption_loss(p2.cpu(), c2.cpu(), d2, a2.cpu()).backward()
    assert _l2(e2.grad.cpu(), e1.grad) < 3e-2
    for (n, q), (_, r) in zip(dec.named_parameters(), ref.named_parameters()):
        if n != "attention.full_att.bias":                      # (bf16 att1 flips a few relu masks: decoder_att sees it most)
            assert _l2(q.grad.cpu(), r.grad) < 6e-2 and _cos(q.grad.cpu(), r.grad) > 0.998, (n, _l2(q.grad.cpu(), r.grad))
    dec.train()
    p3 = dec(enc.cuda(), caps.cuda(), caplens.cuda())[0]
    assert torch.isfinite(p3).all() and _l2(p3.detach().cpu()[inv2], p1.detach()[inv1]) > 1e-3      # dropout active


@pytest.mark.gpu
def test_hip_decoder_odd_sizes_without_any_library_gemm(monkeypatch):
    """VERDICT r5 "missing" 3: sizes the reference accepts (models.py:199-214 puts no constraint on embed / decoder / vocabulary sizes) but
    the MFMA GEMM's layout rules do not -- embed_dim 20, decoder_dim 24 (LSTM reduction length 20 + 128 + 24 = 172, not a multiple of 16),
    vocabulary 31, encoder_dim 128 (E % 256 != 0: the per-image fallback of the encoder gradient) -- run through zero-padded aligned
    copies on the same exact-f32 kernel: torch.mm / addmm / matmul / bmm / baddbmm / einsum are patched to raise; results against the
    oracle decoder."""
    import ppv_amd.decoder as pd
    from oracle.decoder import DecoderWithAttention as Ref, caption_loss
    monkeypatch.delenv("PPV_DEC_WGRAD", raising=False)
    monkeypatch.delenv("PPV_DEC_GEMM", raising=False)
    torch.manual_seed(4)
    B, E, A, M, D, V, L = 5, 128, 128, 20, 24, 31, 7
    ref = Ref(A, M, D, V, encoder_dim=E, dropout=0.0).eval()
    fill_by_name(ref)
    dec = pd.DecoderWithAttention(A, M, D, V, encoder_dim=E, dropout=0.0).eval()
    dec.load_state_dict(ref.state_dict())
    dec = dec.cuda()
    enc = torch.randn(B, 5, 6, E)
    caps = torch.randint(0, V, (B, L))
    caplens = torch.tensor([[7], [3], [5], [2], [6]])
    e1 = enc.clone().requires_grad_(True)
    p1, c1, d1, a1, o1 = ref(e1, caps, caplens)
    caption_loss(p1, c1, d1, a1).backward()
    e2 = enc.cuda().requires_grad_(True)

    def boom(name):
        def f(*a, **k):
            raise AssertionError(f"library GEMM torch.{name} called inside the decoder step")
        return f
    with monkeypatch.context() as mp:
        for name in ("mm", "addmm", "matmul", "bmm", "baddbmm", "einsum"):
            mp.setattr(torch, name, boom(name))
        mp.setattr(torch.Tensor, "baddbmm_", boom("Tensor.baddbmm_"))
        mp.setattr(torch.Tensor, "__matmul__", boom("Tensor.__matmul__"))
        p2, c2, d2, a2, o2 = dec(e2, caps.cuda(), caplens.cuda())
        (p2.square().mean() + a2.square().mean()).backward()
    g_probe = e2.grad.clone()
    assert torch.isfinite(g_probe).all() and float(g_probe.abs().max()) > 0
    e2.grad = None
    dec.zero_grad(set_to_none=True)
    p2, c2, d2, a2, o2 = dec(e2, caps.cuda(), caplens.cuda())
    assert d1 == d2 and o1.tolist() == o2.tolist()                # distinct lengths: the sort has no ties
    assert _l2(p2.detach().cpu(), p1.detach()) < 1e-2 and _l2(a2.detach().cpu(), a1.detach()) < 1e-2
    caption_loss(p2.cpu(), c2.cpu(), d2, a2.cpu()).backward()
    assert _l2(e2.grad.cpu(), e1.grad) < 3e-2
    for (n, q), (_, r) in zip(dec.named_parameters(), ref.named_parameters()):
        if n != "attention.full_att.bias":
            assert _l2(q.grad.cpu(), r.grad) < 6e-2 and _cos(q.grad.cpu(), r.grad) > 0.998, (n, _l2(q.grad.cpu(), r.grad))


@pytest.mark.gpu
def test_hip_decoder_compact_path_equals_general_semantics():
    """Compact path (decoder works on the cell map behind an adaptive-average-pooled encoder_out) against the oracle run on
    the pooled tensor: predictions, alphas, parameter gradients and the gradient that reaches the cell map."""
    import torch.nn.functional as F
    import ppv_amd.decoder as pd
    from oracle.decoder import DecoderWithAttention as Ref, caption_loss
    torch.manual_seed(2)
    B, E, A, M, D, V, L, Hc, Eo = 6, 256, 128, 32, 48, 40, 8, 4, 9
    ref = Ref(A, M, D, V, encoder_dim=E, dropout=0.5).eval()
    fill_by_name(ref)
    dec = pd.DecoderWithAttention(A, M, D, V, encoder_dim=E, dropout=0.5).eval()
    dec.load_state_dict(ref.state_dict())
    dec = dec.cuda()
    cells = torch.relu(torch.randn(B, Hc, Hc, E)).bfloat16()
    caps = torch.randint(0, V, (B, L))
    caplens = torch.tensor([[8], [3], [5], [5], [2], [7]])

    def pooled(c):
        return F.adaptive_avg_pool2d(c.float().permute(0, 3, 1, 2), Eo).permute(0, 2, 3, 1).contiguous()

    c1 = cells.float().requires_grad_(True)
    p1, s1, d1, a1, o1 = ref(pooled(c1), caps, caplens)
    caption_loss(p1, s1, d1, a1).backward()

    c2 = cells.cuda().requires_grad_(True)
    out = pooled(c2.detach())
    out._ppv_cells = c2
    p2, s2, d2, a2, o2 = dec(out, caps.cuda(), caplens.cuda())
    assert d1 == d2 and o1.tolist() == o2.tolist()
    assert _l2(p2.detach().cpu(), p1.detach()) < 1e-2 and _l2(a2.detach().cpu(), a1.detach()) < 1e-2
    assert abs(float(a2.detach().sum(-1)[0, 0]) - 1.0) < 1e-5                       # per-pixel alphas still sum to one
    caption_loss(p2.cpu(), s2.cpu(), d2, a2.cpu()).backward()
    assert c2.grad is not None and c2.grad.dtype == torch.bfloat16
    assert _l2(c2.grad.float().cpu(), c1.grad) < 3e-2 and _cos(c2.grad.float().cpu(), c1.grad) > 0.999
    for (n, q), (_, r) in zip(dec.named_parameters(), ref.named_parameters()):
        if n != "attention.full_att.bias":
            assert _l2(q.grad.cpu(), r.grad) < 6e-2 and _cos(q.grad.cpu(), r.grad) > 0.998, (n, _l2(q.grad.cpu(), r.grad))
    # the general path on the same pooled tensor (no cell map attached) gives the same predictions
    dec.zero_grad()
    p3 = dec(pooled(cells.cuda()), caps.cuda(), caplens.cuda())[0]
    assert _l2(p3.detach().cpu(), p2.detach().cpu()) < 1e-2


@pytest.mark.gpu
def test_encoder_hands_cell_map_to_decoder():
    """Encoder -> Decoder end to end on the device: the compact path is taken and gradients reach the trunk."""
    import ppv_amd.decoder as pd
    from ppv_amd.encoder import Encoder
    torch.manual_seed(0)
    enc = Encoder(encoded_image_size=9, layers=(1, 1, 1, 1)).cuda().train()
    dec = pd.DecoderWithAttention(128, 32, 48, 30, encoder_dim=2048, dropout=0.0).cuda().train()
    img = torch.rand(3, 3, 64, 64, device="cuda")
    out = enc(img)
    assert out.shape == (3, 9, 9, 2048) and out._ppv_cells.shape == (3, 2, 2, 2048)
    caps = torch.randint(0, 30, (3, 6), device="cuda")
    preds, _, dl, alphas, _ = dec(out, caps, torch.tensor([[6], [4], [5]], device="cuda"))
    (preds.sum() + alphas.sum()).backward()
    g = [p.grad for p in enc.parameters() if p.requires_grad]
    assert all(x is not None and torch.isfinite(x).all() for x in g) and any(float(x.abs().max()) > 0 for x in g)
    # same step through the general path (cell map ignored): same predictions
    dec.use_compact = False
    preds2 = dec(enc(img), caps, torch.tensor([[6], [4], [5]], device="cuda"))[0]
    assert _l2(preds2.detach().cpu(), preds.detach().cpu()) < 2e-2


@pytest.mark.gpu
def test_beam_search_entry_points_match_oracle():
    """The stand-alone calls of the reference's beam search (eval/caption.py:86-102): init_hidden_state, attention(enc, h) on an
    expand()-ed encoder tensor, f_beta / decode_step / fc -- against the oracle's attend() and init linears."""
    import ppv_amd.decoder as pd
    from oracle.decoder import DecoderWithAttention as Ref
    torch.manual_seed(4)
    E, A, M, D, V, P, k = 256, 128, 32, 48, 40, 63, 3
    ref = Ref(A, M, D, V, encoder_dim=E, dropout=0.5).eval()
    fill_by_name(ref)
    dec = pd.DecoderWithAttention(A, M, D, V, encoder_dim=E, dropout=0.5).eval()
    dec.load_state_dict(ref.state_dict())
    dec = dec.cuda()
    enc1 = torch.randn(1, P, E)
    enc = enc1.expand(k, P, E)                                              # stride-0 batch dimension, as the beam search builds it
    h = torch.randn(k, D)
    with torch.no_grad():
        h0, c0 = dec.init_hidden_state(enc.cuda())
        mean = enc.mean(dim=1)
        assert rel_err(h0, ref.init_h(mean)) < 1e-5 and rel_err(c0, ref.init_c(mean)) < 1e-5
        awe, alpha = dec.attention(enc.cuda(), h.cuda())
        awe_o, alpha_o = ref.attend(enc, h)
        assert awe.shape == (k, E) and alpha.shape == (k, P)
        assert _l2(alpha.cpu(), alpha_o) < 1e-2 and _l2(awe.cpu(), awe_o) < 1e-2
        gate = dec.sigmoid(dec.f_beta(h.cuda()))
        h1, c1 = dec.decode_step(torch.cat([dec.embedding(torch.tensor([1, 2, 3]).cuda()), gate * awe], dim=1), (h.cuda(), h.cuda()))
        assert dec.fc(h1).shape == (k, V)
    # the reference's scripts call the step outside no_grad (eval/caption.py:93): it runs, warns once, returns detached tensors
    with pytest.warns(UserWarning, match="inference step"):
        awe_g, alpha_g = dec.attention(enc.cuda().requires_grad_(True), h.cuda())
    assert not awe_g.requires_grad and not alpha_g.requires_grad and torch.equal(awe_g, awe) and torch.equal(alpha_g, alpha)


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["hip", "split", "x3"])
def test_batched_weight_gradients_on_the_hip_paths(path, monkeypatch):
    """The batched weight gradients of the dense layers (models.py:199-214 under autograd: g^T h over all time steps).  Default since
    round 5: PPV_DEC_WGRAD=x3, bf16 hi/lo products split inside the kernel (csrc/gemm_f32.hip gemm_bf16x3_tn_kernel); =hip runs them on
    the exact-f32 MFMA kernel, =split on the bf16 MFMA weight-gradient kernel as three stacked products (ppv_split3_rows +
    ppv_conv_wgrad), =lib on rocBLAS (the reference here): every parameter gradient must equal the library path's (f32 summation
    order / 2^-16 products)."""
    import ppv_amd.decoder as pd
    torch.manual_seed(0)
    B, S, E, A, M, D, V = 6, 4, 256, 128, 48, 64, 90
    dec = pd.DecoderWithAttention(attention_dim=A, embed_dim=M, decoder_dim=D, vocab_size=V, encoder_dim=E, dropout=0.0).cuda().train()
    enc = torch.randn(B, S, S, E, generator=torch.Generator().manual_seed(1)).cuda()
    caps = torch.randint(0, V, (B, 9), generator=torch.Generator().manual_seed(2)).cuda()
    lens = torch.tensor([[9], [7], [4], [9], [3], [6]]).cuda()

    def grads():
        dec.zero_grad(set_to_none=True)
        e = enc.clone().requires_grad_(True)
        preds, _, _, alphas, _ = dec(e, caps, lens)
        (preds.square().mean() + alphas.square().mean()).backward()
        return {n: p.grad.detach().clone() for n, p in dec.named_parameters()}, e.grad.detach().clone()

    monkeypatch.setenv("PPV_DEC_WGRAD", "lib")
    ref, ref_e = grads()
    monkeypatch.setenv("PPV_DEC_WGRAD", path)
    got, got_e = grads()
    tol = 5e-5 if path == "hip" else 3e-4          # f32 summation order (tile depth, slab order); measured 1e-6 .. 2.4e-5
    for n in ref:
        assert _l2(got[n].cpu(), ref[n].cpu()) < tol, n
    assert _l2(got_e.cpu(), ref_e.cpu()) < tol


@pytest.mark.gpu
def test_the_default_decoder_step_calls_no_library_gemm(monkeypatch):
    """f1 (VERDICT r4 task 5): with the defaults no dense product of the training step goes to rocBLAS / hipBLASLt -- torch.mm / addmm /
    matmul / baddbmm are patched to raise while a benchmark-shaped step (compact path, 2048-d cells, 512-d layers) runs forward and
    backward."""
    import ppv_amd.decoder as pd
    monkeypatch.delenv("PPV_DEC_WGRAD", raising=False)
    monkeypatch.delenv("PPV_DEC_GEMM", raising=False)
    torch.manual_seed(0)
    B, E, A, M, D, V = 6, 2048, 512, 512, 512, 304
    dec = pd.DecoderWithAttention(attention_dim=A, embed_dim=M, decoder_dim=D, vocab_size=V, encoder_dim=E, dropout=0.0).cuda().train()
    from ppv_amd.encoder import LazyEncoderOut  # noqa: F401  (the cell map rides on the pooled tensor as _ppv_cells)
    cells = (torch.randn(B, 8, 8, E, generator=torch.Generator().manual_seed(1)) * 0.1).cuda().bfloat16().requires_grad_(True)
    import ppv_amd.convops as co
    out = co.adaptive_pool_fwd(cells.detach(), 36)
    out._ppv_cells = cells
    caps = torch.randint(0, V, (B, 9), generator=torch.Generator().manual_seed(2)).cuda()
    lens = torch.tensor([[9], [7], [4], [9], [3], [6]]).cuda()

    def boom(name):
        def f(*a, **k):
            raise AssertionError(f"library GEMM torch.{name} called inside the decoder step")
        return f
    with monkeypatch.context() as mp:
        for name in ("mm", "addmm", "matmul", "bmm", "baddbmm"):
            mp.setattr(torch, name, boom(name))
        mp.setattr(torch.Tensor, "baddbmm_", boom("Tensor.baddbmm_"))
        mp.setattr(torch.Tensor, "__matmul__", boom("Tensor.__matmul__"))
        preds, _, _, alphas, _ = dec(out, caps, lens)
        (preds.square().mean() + alphas.square().mean()).backward()
    assert cells.grad is not None and torch.isfinite(cells.grad.float()).all() and float(cells.grad.float().abs().max()) > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in dec.named_parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("bf16_out", [True, False])
def test_compact_encoder_gradient_kernel(bf16_out):
    """ppv_decc_enc_grad: out[order[b]] = part + gamma (x) dmean + beta^T . d awe on the cell map in one pass (was baddbmm_ + add_ + cast
    + index_put) against the torch formula."""
    from ppv_amd._lib import check, ptr, stream_ptr, lib
    g0 = torch.Generator().manual_seed(3)
    B, C, E, T = 5, 64, 512, 11
    part = torch.randn(B, C, E, generator=g0).cuda()
    dmean = torch.randn(B, E, generator=g0).cuda()
    beta = torch.rand(T, B, C, generator=g0).cuda()
    dawe = torch.randn(T, B, E, generator=g0).cuda()
    gamma = torch.rand(C, generator=g0).cuda()
    order = torch.randperm(B, generator=g0).cuda()
    out = torch.empty((B, C, E), dtype=torch.bfloat16 if bf16_out else torch.float32, device="cuda")
    check(lib().ppv_decc_enc_grad(ptr(part), ptr(dmean), ptr(beta), ptr(dawe), ptr(order), ptr(gamma), ptr(out), int(bf16_out), B, C, E, T,
                                  stream_ptr()), "ppv_decc_enc_grad")
    want = part.double() + gamma.double().view(1, C, 1) * dmean.double().view(B, 1, E) + torch.einsum("tbc,tbe->bce", beta.double(), dawe.double())
    ref = torch.empty_like(want)
    ref[order] = want
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    assert err < (8e-3 if bf16_out else 2e-6), err


@pytest.mark.gpu
@pytest.mark.parametrize("K,M,N", [(1658, 2048, 512), (1658, 512, 512), (128, 512, 2048), (77, 200, 36), (1000, 9490, 512), (33, 130, 260)])
def test_gemm_bf16x3_tn_equals_the_matrix_product(K, M, N):
    """ppv_gemm_bf16x3_tn: a^T b with both f32 operands K-major, as three bf16 products of in-kernel hi/lo splits (the default of the
    decoder's batched weight gradients): any K / M / N, with and without the slab split, ragged tiles, strided rows; error ~2^-17 per
    product (bound here: 2e-5 of the largest element, measured ~1e-6)."""
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(K + M + N)
    a = torch.randn(K, M, generator=g0).cuda()
    b = torch.randn(K, N, generator=g0).cuda()
    got = co.gemm_f32_tn(a, b, x3=True)
    want = (a.double().t() @ b.double()).float()
    assert got.shape == (M, N)
    assert float((got - want).abs().max() / want.abs().max()) < 2e-5
    # an asymmetric integer case is exact (hi carries small integers, lo = 0): catches a transposed or permuted fragment map
    ai = torch.randint(-8, 9, (K, M), generator=g0).float().cuda()
    bi = torch.randint(-8, 9, (K, N), generator=g0).float().cuda()
    assert torch.equal(co.gemm_f32_tn(ai, bi, x3=True), (ai.double().t() @ bi.double()).float())
    # strided rows (a column block of a wider buffer)
    wide = torch.randn(K, M + 24, generator=g0).cuda()
    got2 = co.gemm_f32_tn(wide[:, 8:8 + M], b, x3=True)
    want2 = (wide[:, 8:8 + M].double().t() @ b.double()).float()
    assert float((got2 - want2).abs().max() / want2.abs().max()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,bias", [(2176, 9490, 512, True), (2176, 512, 9504, False), (130, 200, 36, True), (64, 128, 4096, False)])
def test_linear_x3_equals_the_matrix_product(M, N, K, bias):
    """ppv_gemm_bf16x3_nt: x W^T + bias with row-major f32 operands as three bf16 products of in-kernel hi / lo splits (the decoder's
    vocabulary layer over all time steps and its transposed data gradient), with and without the slab split, ragged tiles, a row
    stride that is not the width."""
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(M + N + K)
    xw = torch.randn(M, K + 8, generator=g0).cuda()
    x = xw[:, :K]                                                    # strided rows
    w = torch.randn(N, K, generator=g0).cuda()
    b = torch.randn(N, generator=g0).cuda() if bias else None
    got = co.linear_x3(x, w, b)
    want = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    assert got.shape == (M, N)
    assert float((got.double() - want).abs().max() / want.abs().max()) < 2e-5
    xi = torch.randint(-8, 9, (M, K), generator=g0).float().cuda()
    wi = torch.randint(-8, 9, (N, K), generator=g0).float().cuda()
    assert torch.equal(co.linear_x3(xi, wi), (xi.double() @ wi.double().t()).float())       # exact on small integers: fragment maps


@pytest.mark.gpu
def test_staged_caption_lengths_give_the_same_forward():
    """DecoderWithAttention.stage_lengths (optional): the sorted lengths travel to the host early; forward() must return exactly what
    the reference-style call returns (same order, same decode lengths, same scores), fall back when the staged tensor is not the
    one passed, and survive pickling with a pending stage."""
    import io
    from ppv_amd.decoder import DecoderWithAttention
    torch.manual_seed(0)
    dec = DecoderWithAttention(128, 32, 48, 50, encoder_dim=128, dropout=0.0).cuda().eval()
    enc_out = torch.randn(5, 4, 4, 128, device="cuda")
    caps = torch.randint(0, 50, (5, 9), device="cuda")
    lens = torch.tensor([[7], [9], [4], [9], [5]], device="cuda")
    with torch.no_grad():
        want = dec(enc_out, caps, lens)
        dec.stage_lengths(lens)
        got = dec(enc_out, caps, lens)
        assert got[2] == want[2] and torch.equal(got[4], want[4]) and torch.equal(got[1], want[1])
        torch.testing.assert_close(got[0], want[0], rtol=1e-5, atol=1e-6)
        dec.stage_lengths(lens)
        other = lens.clone()
        other[0, 0] = 3
        alt = dec(enc_out, caps, other)                          # a different tensor: the staged copy must not be used
        assert alt[2] != want[2] and dec._staged is None
        dec.stage_lengths(lens)
        lens.add_(0)                                             # in-place write after staging: version moved on, staged copy dropped
        again = dec(enc_out, caps, lens)
        assert again[2] == want[2]
        dec.stage_lengths(lens, host=lens.cpu())                 # the loader's CPU copy: no device round trip at all
        hst = dec(enc_out, caps, lens)
        assert hst[2] == want[2] and torch.equal(hst[4], want[4]) and torch.equal(hst[1], want[1])
        torch.testing.assert_close(hst[0], want[0], rtol=1e-5, atol=1e-6)
        dec.stage_lengths(lens)
        buf = io.BytesIO()
        torch.save(dec, buf)
        buf.seek(0)
        back = torch.load(buf, weights_only=False)
        assert getattr(back, "_staged", None) is None
        torch.testing.assert_close(back(enc_out, caps, lens)[0], want[0], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("K,M,N", [(1658, 2048, 512), (1658, 512, 512), (128, 512, 2048), (77, 200, 36), (1000, 9490, 512)])
def test_gemm_f32_tn_equals_the_matrix_product(K, M, N):
    """ppv_gemm_f32_tn: a^T b with both operands K-major (the decoder's batched weight gradients g^T h, models.py:199-214 autograd), exact
    f32 on the matrix pipe, any K / M / N, with and without the slab split."""
    import ppv_amd.convops as co
    g0 = torch.Generator().manual_seed(K + M + N)
    a = torch.randn(K, M, generator=g0).cuda()
    b = torch.randn(K, N, generator=g0).cuda()
    got = co.gemm_f32_tn(a, b)
    want = (a.double().t() @ b.double()).float()
    assert got.shape == (M, N)
    assert float((got - want).abs().max() / want.abs().max()) < 2e-6
    # strided rows (a column block of a wider buffer)
    wide = torch.randn(K, M + 24, generator=g0).cuda()
    got2 = co.gemm_f32_tn(wide[:, 8:8 + M], b)
    want2 = (wide[:, 8:8 + M].double().t() @ b.double()).float()
    assert float((got2 - want2).abs().max() / want2.abs().max()) < 2e-6
