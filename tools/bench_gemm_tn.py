"""Micro-benchmark of ppv_gemm_f32_tn on the decoder's batched weight gradients g^T h (K = decode positions of a B = 128 step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
dev = torch.device("cuda", 0)
K = 1658
SH = [("fc (vocab)", 9490, 512), ("lstm W_ih", 2048, 2560), ("lstm W_hh", 2048, 512), ("f_beta", 2048, 512), ("decoder_att", 512, 512), ("init_h (K=128)", 512, 2048)]
th_s = tl_s = t3_s = 0
for name, M, N in SH:
    k = 128 if "K=128" in name else K
    a = torch.randn(k, M, device=dev); b = torch.randn(k, N, device=dev)
    def timeit(fn, n=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    th = timeit(lambda: co.gemm_f32_tn(a, b)); tl = timeit(lambda: a.t() @ b); t3 = timeit(lambda: co.gemm_f32_tn(a, b, x3=True))
    fl = 2.0 * k * M * N
    print(f"{name:16s} K{k:5d} M{M:5d} N{N:5d}: exact-f32 hip {th:7.1f} us ({fl/th/1e6:5.1f} TF)  lib {tl:7.1f} us ({fl/tl/1e6:5.1f} TF)  bf16x3 {t3:7.1f} us ({fl/t3/1e6:5.1f} TF eff.)", flush=True)
    th_s += th; tl_s += tl; t3_s += t3
print(f"sum exact-f32 hip {th_s:.1f}  lib {tl_s:.1f}  bf16x3 {t3_s:.1f}")
