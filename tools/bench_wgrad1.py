"""The 1x1 weight-gradient shapes of the trunk alone (cold rotating operands), for kernel-trace runs:
rocprofv3 --kernel-trace --stats -- python tools/bench_wgrad1.py [variant]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
B, NB = 128, 5
v = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
co.L().ppv_wgrad_set_variant(v)
for cin, cout, h in [(256, 1024, 16), (1024, 256, 16), (128, 512, 32), (512, 128, 32), (512, 2048, 8), (2048, 512, 8)]:
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    gs = [torch.randn(B, h, h, cout, device="cuda").bfloat16() for _ in range(NB)]
    acc = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
    for i in range(6 * NB):
        co.conv_wgrad(gs[i % NB], xs[i % NB], 1, 1, 1, 0, scratch=acc)
    torch.cuda.synchronize()
co.L().ppv_wgrad_set_variant(0)
