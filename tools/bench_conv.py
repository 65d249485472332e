"""Micro-benchmark of the conv GEMM kernel variants on the ResNet-101 shapes at B images."""
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
    ho = h // st
    part = torch.zeros(co.stat_tiles(B * ho * ho), 2, cout, device="cuda")
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    line = f"cin {cin:5d} cout {cout:5d} k{k} s{st} h{h:3d}:"
    for variant in (1, 3, 4):
        if cout % 128 and variant > 1:
            continue
        co.L().ppv_conv_set_variant(variant)
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
        line += f"  v{variant} {ms*1e3:7.1f} us {fl/ms/1e9:6.0f} TF"
    print(line)
co.L().ppv_conv_set_variant(0)
