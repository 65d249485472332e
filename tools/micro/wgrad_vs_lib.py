"""Reference ceiling for the 1x1 weight gradient (cdna_hip_programming.md 5.4 rule 10: a ceiling claim needs a known-good reference on
the same hardware): dW[Cout, Cin] = G^T X is a plain GEMM with K = B*H*W; time hipBLASLt (torch.mm, bf16 in / f32 accumulate) on the
benchmark's shapes with cold (rotating) operands next to ppv_conv_wgrad.  Measurement only -- the product never calls the library here."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ppv_amd.convops as co

B, NB = 128, 5
for cin, cout, h in [(256, 1024, 16), (1024, 256, 16), (128, 512, 32), (512, 128, 32), (512, 2048, 8), (2048, 512, 8)]:
    M = B * h * h
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    gs = [torch.randn(B, h, h, cout, device="cuda").bfloat16() for _ in range(NB)]
    scratch = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")

    def t(fn, n=4 * NB):
        for i in range(NB):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i % NB)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    ours = t(lambda i: co.conv_wgrad(gs[i], xs[i], 1, 1, 1, 0, scratch=scratch))
    lib_bf16 = t(lambda i: torch.mm(gs[i].view(M, cout).t(), xs[i].view(M, cin)))
    try:
        lib_f32 = t(lambda i: torch.mm(gs[i].view(M, cout).t(), xs[i].view(M, cin), out_dtype=torch.float32))
    except Exception as e:  # noqa: BLE001
        lib_f32 = float("nan")
    fl = 2.0 * M * cin * cout
    print(f"cin {cin:5d} cout {cout:5d} h {h:3d}  M {M:6d}:  ppv_conv_wgrad {ours:6.1f} us {fl/ours/1e6:5.0f} TF | hipBLASLt bf16-out {lib_bf16:6.1f} us {fl/lib_bf16/1e6:5.0f} TF | f32-out {lib_f32:6.1f} us",
          flush=True)
