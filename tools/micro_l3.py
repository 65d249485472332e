"""Layer-3 micro workload for PMC profiling: conv fwd 1x1/3x3 + wgrad, cold operands (rotating buffers > 256 MiB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
B, h = 128, 16
NB = 6
xs256 = [torch.randn(B, h, h, 256, device="cuda").bfloat16() for _ in range(NB)]
xs1024 = [torch.randn(B, h, h, 1024, device="cuda").bfloat16() for _ in range(NB)]
w1 = co.weight_layout(torch.randn(1024, 256, 1, 1, device="cuda") * 0.05, 0)
w2 = co.weight_layout(torch.randn(256, 1024, 1, 1, device="cuda") * 0.05, 0)
w3 = co.weight_layout(torch.randn(256, 256, 3, 3, device="cuda") * 0.05, 0)
part = torch.zeros(32, 2, 1024, device="cuda")
for it in range(3 * NB):
    i = it % NB
    co.conv_fwd(xs256[i], w1, 1, 0, stat_part=part)          # K=256  N=1024
    co.conv_fwd(xs1024[i], w2, 1, 0, stat_part=part)         # K=1024 N=256
    co.conv_fwd(xs256[i], w3, 1, 1, stat_part=part)          # K=2304 N=256
    co.conv_wgrad(xs1024[i], xs256[(i + 1) % NB], 1, 1, 1, 0)     # dW 1024x256
    co.conv_wgrad(xs256[i], xs1024[(i + 1) % NB], 1, 1, 1, 0)     # dW 256x1024
    co.conv_wgrad(xs256[i], xs256[(i + 1) % NB], 3, 3, 1, 1)      # dW 256x256x3x3
torch.cuda.synchronize()
