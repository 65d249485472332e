"""Layer-1 convolutions on the 128 x 64 tile (B = 128, 64 x 64 maps), cold rotating operands: forward with BatchNorm statistics
and data gradient with the fused BN-backward sums.  Run on the GPU box: python tools/bench_layer1.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co

B, h = 128, 64
M = B * h * h


def timed(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, k in [(64, 64, 3), (256, 64, 1), (64, 64, 1)]:
    NB = 6
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    gs = [torch.randn(B, h, h, cout, device="cuda").bfloat16() for _ in range(NB)]
    xr = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(3)]
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.02
    wf, wd = co.weight_layout(w, 0), co.weight_layout(w, 1)
    part = torch.zeros(co.stat_tiles(M), 2, cout, device="cuda")
    pr = torch.zeros(64 * cin, device="cuda")
    coef = torch.rand(4, cin, device="cuda") + 0.5
    it = [0]
    pad = (k - 1) // 2

    def fwd():
        it[0] += 1
        return co.conv_fwd(xs[it[0] % NB], wf, 1, pad, stat_part=part)

    def dg():
        it[0] += 1
        return co.conv_dgrad(gs[it[0] % NB], wd, 1, pad, (h, h), red=(xr[it[0] % 3], pr, coef))
    by = M * (cin + cout) * 2
    for nm, fn, extra in (("fwd+stats", fwd, 0), ("dgrad+mask+sums", dg, M * cin * 2)):
        t = timed(fn)
        print(f"{k}x{k} {cin}->{cout} @{h}x{h} {nm}: {t:6.1f} us  ({(by + extra) / t / 1e6:5.2f} TB/s of compulsory bytes)")
