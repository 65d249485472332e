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
