"""conv2d_f32(exact=True) -- forward, data gradient, weight gradient -- against torch CPU f64 on the trunk's layer kinds (relative max-norm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ppv_amd.nn_ops import conv2d_f32

rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
g = torch.Generator().manual_seed(0)
import itertools
extra = [(4, 8, 256, 256, 3, 2, 1), (4, 8, 512, 256, 1, 1, 0), (4, 4, 256, 1024, 1, 1, 0), (4, 8, 512, 1024, 1, 2, 0), (4, 4, 1024, 512, 1, 1, 0), (4, 4, 512, 512, 3, 2, 1), (4, 32, 64, 64, 3, 1, 1)]
for (B, H, Cin, Cout, k, s, p) in extra + [(4, 64, 8, 64, 7, 2, 3), (4, 16, 64, 64, 1, 1, 0), (4, 16, 64, 64, 3, 1, 1), (4, 16, 64, 256, 1, 1, 0), (4, 16, 256, 128, 1, 1, 0),
                                   (4, 16, 128, 128, 3, 2, 1), (4, 16, 256, 512, 1, 2, 0), (4, 4, 512, 512, 3, 1, 1), (4, 2, 2048, 512, 1, 1, 0)]:
    x = torch.randn(B, Cin, H, H, generator=g).double()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).double()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=s, padding=p)
    gy = torch.randn(yr.shape, generator=g).double()
    yr.backward(gy)
    xg = x.float().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    wg = torch.nn.Parameter(w.float().cuda())
    y = conv2d_f32(xg, wg, None, s, p, weight_grad=True, accurate_wgrad=True, exact=True)
    y.backward(gy.float().permute(0, 2, 3, 1).contiguous().cuda())
    print(f"B{B} H{H} {Cin}->{Cout} k{k} s{s}: fwd {rel(y.detach().permute(0, 3, 1, 2), yr.detach()):.1e}  dgrad {rel(xg.grad.permute(0, 3, 1, 2), xr.grad):.1e}  "
          f"wgrad {rel(wg.grad, wr.grad):.1e}", flush=True)
