"""Micro-benchmark of the conv GEMM kernel on the ResNet-101 shapes at B images."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(64, 64, 1, 1, 64), (64, 64, 3, 1, 64), (64, 256, 1, 1, 64), (256, 64, 1, 1, 64), (128, 128, 3, 1, 32),
          (128, 512, 1, 1, 32), (512, 128, 1, 1, 32), (256, 256, 3, 1, 16), (256, 1024, 1, 1, 16), (1024, 256, 1, 1, 16),
          (512, 512, 3, 1, 8), (512, 2048, 1, 1, 8), (2048, 512, 1, 1, 8), (128, 128, 3, 2, 64), (256, 512, 1, 2, 64)]
for cin, cout, k, st, h in SHAPES:
    x = torch.randn(B, h, h, cin, device="cuda").bfloat16()
    w = co.weight_layout(torch.randn(cout, cin, k, k, device="cuda") * 0.05, 0)
    ho_ = h // st; part = torch.zeros(co.stat_tiles(B * ho_ * ho_), 2, cout, device="cuda")
    for _ in range(3):
        y = co.conv_fwd(x, w, st, (k - 1) // 2, stat_part=part)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        y = co.conv_fwd(x, w, st, (k - 1) // 2, stat_part=part)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    ho = h // st
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    byts = (x.numel() + y.numel() + w.numel()) * 2
    print(f"cin {cin:5d} cout {cout:5d} k{k} s{st} h{h:3d}: {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TF/s  {byts/ms/1e6:8.1f} GB/s")
