"""Deterministic ResNet-101 trunk parameters / buffers keyed on state_dict names (a 170 MB weight fixture would not fit the
repository): every tensor is drawn from a generator seeded with the CRC of its name, so the reference's Encoder in
tests/golden/make_golden.py::gen_encoder, the CPU oracle and the MI355X trunk hold the same numbers without a file."""
import zlib

import torch


def fill_trunk_by_name(module):
    with torch.no_grad():
        for name, t in module.state_dict().items():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if name.endswith("num_batches_tracked"):
                continue
            if name.endswith("running_var"):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
            elif name.endswith("running_mean"):
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)
            elif t.dim() == 1:                                     # BatchNorm gamma / beta
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75 if name.endswith("weight") else torch.randn(t.shape, generator=g) * 0.1)
            else:                                                  # convolution: He fan-out, as torchvision initialises it
                fan_out = t.shape[0] * t.shape[2] * t.shape[3]
                t.copy_(torch.randn(t.shape, generator=g) * (2.0 / fan_out) ** 0.5)
