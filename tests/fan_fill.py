"""Deterministic FAN parameters keyed on state_dict names (same procedure as tests/golden/make_golden.py::fill_by_name)."""
import zlib

import torch


def fill_by_name(module):
    with torch.no_grad():
        for name, t in module.state_dict().items():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if name.endswith("num_batches_tracked"):
                continue
            if name.endswith("running_var"):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
            elif name.endswith("running_mean"):
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)
            elif t.dim() == 1 and ("bn" in name or "downsample.0" in name):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75 if name.endswith("weight") else torch.randn(t.shape, generator=g) * 0.1)
            elif t.dim() == 1:
                t.copy_(torch.randn(t.shape, generator=g) * 0.05)
            else:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * (2.0 / fan_in) ** 0.5 * (0.004 if name.startswith("l0.") else 1.0))
