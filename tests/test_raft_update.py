"""RAFT's SepConvGRU (reference Face-DeId/RAFT/core/update.py:33-60): the oracle against the golden produced by the
reference module itself (tests/golden/make_golden.py raft_gru), and the HIP module against both, three chained updates."""
import zlib

import pytest
import torch

from conftest import load_golden, rel_err


def _params():
    shapes = {}
    for tag, k in (("1", (1, 5)), ("2", (5, 1))):
        for g in "zrq":
            shapes[f"conv{g}{tag}.weight"] = (128, 384) + k
            shapes[f"conv{g}{tag}.bias"] = (128,)
    out = {}
    for name, shape in shapes.items():                                   # the generator's fill, keyed on the state_dict name
        gen = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        t = torch.randn(shape, generator=gen)
        out[name] = t * ((1.0 / t[0].numel()) ** 0.5 if t.dim() > 1 else 0.1)
    return out


def test_oracle_matches_the_reference_module():
    from oracle.raft_update import sep_conv_gru
    g = load_golden("raft_gru.npz")
    p = _params()
    h, x = torch.from_numpy(g["h"]), torch.from_numpy(g["x"])
    h1 = sep_conv_gru(h, x, p)
    assert rel_err(h1, g["out1"]) < 1e-5
    h3 = sep_conv_gru(sep_conv_gru(h1, x, p), x, p)
    assert rel_err(h3, g["out3"]) < 1e-5


@pytest.mark.gpu
def test_hip_sepconvgru_matches_the_reference_golden():
    from ppv_amd.raft_update import SepConvGRU
    g = load_golden("raft_gru.npz")
    gru = SepConvGRU(hidden_dim=128, input_dim=256)
    assert sorted(gru.state_dict().keys()) == sorted(_params().keys())     # the reference's parameter names
    gru.load_state_dict(_params())
    gru = gru.cuda()
    h, x = torch.from_numpy(g["h"]).cuda(), torch.from_numpy(g["x"]).cuda()
    h1 = gru(h, x)                                                         # called outside no_grad, as RAFT's loop does
    assert h1.shape == h.shape and not h1.requires_grad
    assert rel_err(h1, g["out1"]) < 1e-3
    h3 = gru(gru(h1, x), x)
    assert rel_err(h3, g["out3"]) < 1e-3                                   # north_star tolerance, after three recurrences
