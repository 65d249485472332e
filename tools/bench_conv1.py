"""The forward convolution shapes that carry the step (cold rotating operands), for kernel-trace runs:
rocprofv3 --kernel-trace --stats -- python tools/bench_conv1.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
B = 128
for cin, cout, k, h in [(256, 1024, 1, 16), (1024, 256, 1, 16), (256, 256, 3, 16), (128, 512, 1, 32), (512, 128, 1, 32), (512, 2048, 1, 8), (2048, 512, 1, 8)]:
    NB = max(2, int(600e6 // (B * h * h * (cin + cout) * 2)) + 1)
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    w = co.weight_layout(torch.randn(cout, cin, k, k, device="cuda") * 0.05, 0)
    part = torch.zeros(co.stat_tiles(B * h * h), 2, cout, device="cuda")
    for i in range(5 * NB):
        co.conv_fwd(xs[i % NB], w, 1, (k - 1) // 2, stat_part=part)
    torch.cuda.synchronize()
