"""Pins the trunk oracle (oracle/resnet.py Encoder) against tests/golden/encoder.npz: the output of the REFERENCE's own
``models.Encoder`` class (Image_Caption/models.py:8-54) run on CPU by tests/golden/make_golden.py::gen_encoder -- children()[:-2],
fine_tune() on children [5:], AdaptiveAvgPool2d + permute, train-mode BatchNorm (train.py:245).  CPU only."""
import numpy as np
import torch

from conftest import load_golden, rel_err
from trunk_fill import fill_trunk_by_name


def _oracle(E):
    from oracle.resnet import Encoder
    enc = Encoder(encoded_image_size=E)
    fill_trunk_by_name(enc)
    return enc.train()


def test_state_dict_keys_and_trainable_set_are_the_references():
    g = load_golden("encoder.npz")
    enc = _oracle(3)
    assert list(enc.state_dict().keys()) == [str(k) for k in g["state_names"]]
    assert [n for n, _ in enc.named_parameters()] == [str(k) for k in g["param_names"]]
    assert [p.requires_grad for _, p in enc.named_parameters()] == [bool(b) for b in g["requires_grad"]]
    # models.py:43-54: stem + layer1 frozen, layer2-4 trainable
    assert not any(p.requires_grad for n, p in enc.named_parameters() if n.split(".")[1] in "014")
    assert all(p.requires_grad for n, p in enc.named_parameters() if n.split(".")[1] in "567")


def test_forward_backward_and_running_statistics_match_the_reference_encoder():
    g = load_golden("encoder.npz")
    enc = _oracle(3)
    img = torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(0)).requires_grad_(True)
    y = enc(img)
    assert y.shape == (4, 3, 3, 2048)
    assert rel_err(y.detach(), g["out"]) < 1e-5
    w = torch.rand(y.shape, generator=torch.Generator().manual_seed(5))
    (y * w).sum().backward()
    assert rel_err(img.grad, g["img_grad"]) < 1e-4
    gs = g["param_grad_stats"]
    for i, (n, p) in enumerate(enc.named_parameters()):
        if p.grad is None:
            assert not gs[i].any(), n
            continue
        v = p.grad.double().reshape(-1)
        assert abs((v * v).sum().item() - gs[i, 1]) <= 1e-3 * gs[i, 1] + 1e-30, n
        k = min(8, v.numel())
        assert np.abs(v[:k].numpy() - gs[i, 2:2 + k]).max() <= 1e-3 * np.sqrt(gs[i, 1] / v.numel()) + 1e-3 * np.abs(gs[i, 2:2 + k]).max(), n
    assert rel_err(enc.resnet[5][0].conv1.weight.grad, g["g_layer2_conv1"]) < 1e-4
    assert rel_err(enc.resnet[7][2].bn3.weight.grad, g["g_layer4_bn3"]) < 1e-4
    sd = enc.state_dict()
    assert rel_err(torch.cat([sd[k] for k in sd if k.endswith("running_mean")]), g["running_mean"]) < 1e-5
    assert rel_err(torch.cat([sd[k] for k in sd if k.endswith("running_var")]), g["running_var"]) < 1e-5
    assert [int(sd[k]) for k in sd if k.endswith("num_batches_tracked")] == list(g["num_batches_tracked"])


def test_default_pool_size_36():
    g = load_golden("encoder.npz")
    enc = _oracle(36)
    with torch.no_grad():
        y = enc(torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(0)))
    assert y.shape == (4, 36, 36, 2048)
    assert rel_err(y[:, ::7, ::7, ::16], g["out36_sub"]) < 1e-5
    s = g["out36_stats"]
    assert abs(y.double().sum().item() - s[0]) < 1e-5 * abs(s[0]) and abs(y.max().item() - s[2]) < 1e-5 * s[2]


def test_product_encoder_surface_equals_the_reference_encoders():
    """b-3: ppv_amd.encoder.Encoder holds the same state_dict keys / parameter order / trainable set as the reference class
    (constructed on CPU: parameter holders only, no kernel is launched)."""
    import ppv_amd  # noqa: F401
    from ppv_amd.encoder import Encoder
    g = load_golden("encoder.npz")
    enc = Encoder(3)
    assert list(enc.state_dict().keys()) == [str(k) for k in g["state_names"]]
    assert [n for n, _ in enc.named_parameters()] == [str(k) for k in g["param_names"]]
    assert [p.requires_grad for _, p in enc.named_parameters()] == [bool(b) for b in g["requires_grad"]]
    enc.fine_tune(False)
    assert not any(p.requires_grad for p in enc.parameters())
