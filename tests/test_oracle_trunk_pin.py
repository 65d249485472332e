"""Pins the trunk oracle (oracle/resnet.py: a torch.nn restatement of torchvision's ResNet-101 v1.5 trunk, models.py:17-21; torchvision
itself is absent offline) against an INDEPENDENT implementation that is present: Hugging Face `transformers` ResNetModel configured as
ResNet-101 v1.5 (bottleneck, stride on the 3x3, depths 3-4-23-3).  Same weights in both -> same activations, eval and train mode (batch
statistics and running-statistics update), and the same gradient of a scalar to the input."""
import pytest
import torch

transformers = pytest.importorskip("transformers")


def _hf_key(k):
    parts = k.split(".")
    if parts[0] == "0":
        return "embedder.embedder.convolution." + parts[1]
    if parts[0] == "1":
        return "embedder.embedder.normalization." + parts[1]
    stage, blk, name = int(parts[0]) - 4, parts[1], parts[2]
    base = f"encoder.stages.{stage}.layers.{blk}."
    if name == "downsample":
        return base + "shortcut." + ("convolution." if parts[3] == "0" else "normalization.") + parts[4]
    idx = int(name[-1]) - 1
    return base + f"layer.{idx}." + ("convolution." if name.startswith("conv") else "normalization.") + parts[3]


def _pair(seed=0):
    from transformers import ResNetConfig, ResNetModel
    from oracle import resnet as R
    torch.manual_seed(seed)
    ours = R.make_resnet101_trunk()
    with torch.no_grad():                                    # non-trivial BatchNorm state
        for m in ours.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 23, 3], layer_type="bottleneck",
                       hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    hf = ResNetModel(cfg)
    sd = {_hf_key(k): v.clone() for k, v in ours.state_dict().items()}
    assert set(sd) == set(hf.state_dict())
    hf.load_state_dict(sd)
    return ours.double(), hf.double()


@pytest.mark.parametrize("train", [False, True])
def test_trunk_oracle_equals_independent_resnet101(train):
    ours, hf = _pair()
    ours.train(train); hf.train(train)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = ours(xa)
    yb = hf(xb).last_hidden_state
    assert ya.shape == yb.shape == (2, 2048, 2, 2)
    assert torch.allclose(ya, yb, rtol=1e-9, atol=1e-9)
    w = torch.randn(ya.shape, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    (ya * w).sum().backward(); (yb * w).sum().backward()
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-7, atol=1e-10)
    if train:                                                # the running statistics moved the same way
        sa, sb = ours.state_dict(), hf.state_dict()
        for k in sa:
            if "running" in k:
                assert torch.allclose(sa[k], sb[_hf_key(k)], rtol=1e-12, atol=1e-12), k
