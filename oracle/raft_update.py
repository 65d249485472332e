"""CPU oracle (torch fp32) for RAFT's separable convolutional GRU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates reference Face-DeId/RAFT/core/update.py:33-60 (SepConvGRU): a horizontal
(1 x 5) and a vertical (5 x 1) GRU update, each  z = sigmoid(conv_z([h, x])), r = sigmoid(conv_r([h, x])),
q = tanh(conv_q([r h, x])), h = (1 - z) h + z q.  Pinned by tests/golden/raft_gru.npz (the reference module itself, run by
tests/golden/make_golden.py)."""
import torch
import torch.nn.functional as F


def sep_conv_gru(h, x, params):
    """params: dict of the six convolutions' weights and biases under the reference's names (convz1.weight ...)."""
    for tag, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(F.conv2d(hx, params[f"convz{tag}.weight"], params[f"convz{tag}.bias"], padding=pad))
        r = torch.sigmoid(F.conv2d(hx, params[f"convr{tag}.weight"], params[f"convr{tag}.bias"], padding=pad))
        q = torch.tanh(F.conv2d(torch.cat([r * h, x], dim=1), params[f"convq{tag}.weight"], params[f"convq{tag}.bias"], padding=pad))
        h = (1 - z) * h + z * q
    return h
