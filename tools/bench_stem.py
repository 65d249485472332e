"""Stem conv forward (resnet.0) at the headline shape, cold-ish: PPV_STEM_ROWS=0/1 A/B (read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
B, H = 128, 256
imgs = [torch.rand(B, 3, H, H, device="cuda") for _ in range(3)]
w = co.stem_weight_layout(torch.randn(64, 3, 7, 7, device="cuda") * 0.1, 0)
part = torch.zeros(co.stat_tiles(B * (H // 2) ** 2), 2, 64, device="cuda")
for i in range(3):
    co.stem_conv(imgs[i], w, part)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 12
e0.record()
for i in range(n):
    co.stem_conv(imgs[i % 3], w, part)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"stem conv B={B} {H}x{H}: {us:.1f} us  ({(B * 3 * H * H * 4 + B * H * H // 4 * 64 * 2) / us / 1e6:.2f} TB/s of compulsory traffic)")

gs = [torch.randn(B, H // 2, H // 2, 64, device="cuda").bfloat16() for _ in range(3)]
wd = co.stem_weight_layout(torch.randn(64, 3, 7, 7, device="cuda") * 0.1, 1)
for i in range(3):
    co.stem_dgrad(gs[i], wd)
torch.cuda.synchronize()
e0.record()
for i in range(n):
    co.stem_dgrad(gs[i % 3], wd)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"stem dgrad: {us:.1f} us  ({(B * 3 * H * H * 4 + B * H * H // 4 * 64 * 2) / us / 1e6:.2f} TB/s of compulsory traffic)")
