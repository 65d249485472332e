"""AdaptiveAvgPool2d(36) on the 8x8 map at B = 128: forward (1.36 GB f32 out) and backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
x = torch.relu(torch.randn(128, 8, 8, 2048, device="cuda")).bfloat16()
for name, fn in (("fwd", lambda: co.adaptive_pool_fwd(x, 36)),):
    for _ in range(3): y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = fn()
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1) / 10 * 1e3, "us")
g = torch.randn_like(y)
for _ in range(3): co.adaptive_pool_bwd(g, (8, 8), relu_of=x)
torch.cuda.synchronize()
e0.record()
for _ in range(10): co.adaptive_pool_bwd(g, (8, 8), relu_of=x)
e1.record(); torch.cuda.synchronize()
print("bwd", e0.elapsed_time(e1) / 10 * 1e3, "us")
