"""Micro-benchmark of ppv_gemm_f32 (csrc/gemm_f32.hip) on the caption decoder's per-time-step shapes (Image_Caption/models.py:199-214),
against torch.addmm (rocBLAS) for reference.  B = 128 rows per step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co

dev = torch.device("cuda", 0)
SH = [("lstm gates fwd", 128, 2048, 3072), ("lstm gates bwd (d input)", 128, 3072, 2048), ("decoder_att", 128, 512, 512),
      ("f_beta fwd", 128, 2048, 512), ("f_beta bwd", 128, 512, 2048), ("vocab fwd", 1664, 9504, 512), ("vocab bwd", 1664, 512, 9504),
      ("h0/c0 init", 128, 512, 2048)]
tot_h = tot_l = 0.0
for name, M, N, K in SH:
    x = torch.randn(M, K, device=dev)
    ws = [torch.randn(N, K, device=dev) * 0.02 for _ in range(4)]
    b = torch.randn(N, device=dev)
    def timeit(fn, n=40):
        for i in range(4):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    out = torch.empty(M, N, device=dev)
    th = timeit(lambda i: co.linear_f32(x, ws[i % 4], b, out=out))
    tl = timeit(lambda i: torch.addmm(b, x, ws[i % 4].t(), out=out))
    err = ((co.linear_f32(x, ws[0], b) - torch.addmm(b, x, ws[0].t())).abs().max() / torch.addmm(b, x, ws[0].t()).abs().max()).item()
    fl = 2.0 * M * N * K
    print(f"{name:26s} M{M:5d} N{N:5d} K{K:5d}: hip {th:7.1f} us ({fl/th/1e6:5.1f} TF, ks {co.L().ppv_gemm_f32_ksplit(M, N, K)})   lib {tl:7.1f} us ({fl/tl/1e6:5.1f} TF)   rel err {err:.1e}")
    tot_h += th; tot_l += tl
print(f"sum: hip {tot_h:.1f} us, lib {tot_l:.1f} us")
