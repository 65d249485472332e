"""vocab-sized exact-f32 GEMM only (for counter passes): M 1664, N 9504, K 512, 20 launches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ppv_amd.convops as co
x = torch.randn(1664, 512, device="cuda"); w = torch.randn(9504, 512, device="cuda") * 0.02; out = torch.empty(1664, 9504, device="cuda")
for _ in range(20):
    co.linear_f32(x, w, None, out=out)
torch.cuda.synchronize()
