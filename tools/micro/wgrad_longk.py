"""What would a slab-free (grouped, un-split) weight gradient reach?  The existing kernel on a 23x longer reduction (23 layer-3 blocks'
rows in one problem): same tiles, same staging, slab + reduce cost amortised 23x -> time / 23 = the per-conv time of the steady state."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ppv_amd.convops as co


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


co.zero_page(torch.device("cuda", 0))
for cin, cout in [(1024, 256), (256, 1024)]:
    for reps in (1, 23):
        B = 128 * reps
        g = torch.randn(B, 16, 16, cout, device="cuda").bfloat16()
        x = torch.randn(B, 16, 16, cin, device="cuda").bfloat16()
        t = timed(lambda: co.conv_wgrad(g, x, 1, 1, 1, 0))
        fl = 2 * B * 256 * cin * cout
        print(f"wgrad 1x1 {cin}->{cout} rows x{reps}: {t:8.1f} us total, {t / reps:6.1f} us per 32768 rows ({fl / t / 1e6:4.0f} TF/s)")
        del g, x
