"""Adaptive pool 8 x 8 -> 36 x 36 at B = 128 (the dense [B,36,36,2048] f32 surface of models.py:39-41 and its gradient): us per call and TB/s,
the per-row / per-pixel kernels of round 6 against the flat-index ones (PPV_POOL_PX=0 in a second process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
B = 128
x = torch.randn(B, 8, 8, 2048, device="cuda").relu().bfloat16()
gy = torch.randn(B, 36, 36, 2048, device="cuda")
out = torch.empty(B, 36, 36, 2048, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
f = t(lambda: co.adaptive_pool_fwd(x, 36, out=out))
b = t(lambda: co.adaptive_pool_bwd(gy, (8, 8), relu_of=x))
gb = out.numel() * 4 / 1e12
print(f"PPV_POOL_PX={os.environ.get('PPV_POOL_PX', '1')}: forward {f:.1f} us ({gb / f * 1e6:.2f} TB/s written), backward {b:.1f} us ({gb / b * 1e6:.2f} TB/s read)")
