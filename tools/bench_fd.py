"""BASELINE.json config 4: Face-DeId Camera + FAN heat-map regressor + RAFT correlation volume, batch 32 @ 512 x 512,
one MI355X (forward, as the reference's Solver.train uses them: solver.py:144-147, core/utils.py:437-462)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd  # noqa: F401
from ppv_amd.camera_optics import Camera
from ppv_amd.fan import FAN
from ppv_amd.raft_corr import CorrBlock, AlternateCorrBlock

dev = torch.device("cuda", 0)
B = 32
x = (torch.rand(B, 3, 512, 512, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
cam = Camera(device=dev, N=512, zernike_terms=300).eval()
fan = FAN().to(dev).eval()
g = torch.Generator().manual_seed(1)
f1 = torch.randn(B, 256, 64, 64, generator=g).to(dev)
f2 = torch.randn(B, 256, 64, 64, generator=g).to(dev)
ys, xs = torch.meshgrid(torch.arange(64), torch.arange(64), indexing="ij")
coords = (torch.stack([xs, ys], 0).float()[None] + 2.0 * torch.randn(B, 2, 64, 64, generator=g)).to(dev)


def timeit(fn, n=5, w=2):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def corr_all():
    for b in range(B):                       # the reference runs RAFT per sample (core/utils.py:452-458)
        blk = CorrBlock(f1[b:b + 1], f2[b:b + 1])
        for _ in range(20):                  # iters=20 lookups
            blk(coords[b:b + 1])


def alt_all():
    for b in range(B):                       # same loop with the on-the-fly correlation (no 64 MB volume per sample)
        blk = AlternateCorrBlock(f1[b:b + 1], f2[b:b + 1])
        for _ in range(20):
            blk(coords[b:b + 1])


t_cam = timeit(lambda: cam(x))
xs_ = cam(x)
t_fan = timeit(lambda: fan.get_heatmap(xs_, Privacy=True))
fan16 = FAN(precision="bf16").to(dev).eval()
t_fan16 = timeit(lambda: fan16.get_heatmap(xs_, Privacy=True))
t_corr = timeit(corr_all, n=2, w=1)
t_alt = timeit(alt_all, n=2, w=1)
print(json.dumps({"config": "FD Camera + FAN + RAFT CorrBlock, B=32 @512x512 (BASELINE.json configs[3]), forward",
                  "camera_ms": round(t_cam * 1e3, 3), "fan_ms": round(t_fan * 1e3, 3), "fan_bf16_mode_ms": round(t_fan16 * 1e3, 3), "corr_32x(volume+20 lookups)_ms": round(t_corr * 1e3, 3),
                  "altcorr_32x(20 on-the-fly lookups)_ms": round(t_alt * 1e3, 3),
                  "images_per_s_camera_fan": round(B / (t_cam + t_fan), 1)}))
