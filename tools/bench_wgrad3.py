"""The 3x3 weight-gradient shapes of the trunk alone (cold rotating operands), for kernel-trace runs:
rocprofv3 --kernel-trace --stats -- python tools/bench_wgrad3.py [variant]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
B, NB = 128, 5
v = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
co.L().ppv_wgrad_set_variant(v)
for c, h in [(256, 16), (128, 32), (512, 8)]:
    xs = [torch.randn(B, h, h, c, device="cuda").bfloat16() for _ in range(NB)]
    gs = [torch.randn(B, h, h, c, device="cuda").bfloat16() for _ in range(NB)]
    acc = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
    for i in range(6 * NB):
        co.conv_wgrad(gs[i % NB], xs[i % NB], 3, 3, 1, 1, scratch=acc)
    torch.cuda.synchronize()
co.L().ppv_wgrad_set_variant(0)
