"""A/B of conv_halo.hip (variant 9) against the tiled kernel (variant 7) on the trunk's 3x3 / stride-1 shapes at B = 128: forward with
BatchNorm statistics and data gradient with the fused BN-backward sums, cold operands.  Run on the GPU box: python tools/bench_halo.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co

B = 128
lib = co.L()


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


for c, h in [(256, 16), (128, 32)]:
    xs = [torch.randn(B, h, h, c, device="cuda").bfloat16() for _ in range(6)]
    xr = [torch.randn(B, h, h, c, device="cuda").bfloat16() for _ in range(3)]
    w = torch.randn(c, c, 3, 3, device="cuda") * 0.02
    wf, wd = co.weight_layout(w, 0), co.weight_layout(w, 1)
    M = B * h * h
    part = torch.zeros(co.stat_tiles(M), 2, c, device="cuda")
    pr = torch.zeros(64 * c, device="cuda")
    coef = torch.rand(4, c, device="cuda") + 0.5
    it = [0]

    def fwd():
        it[0] += 1
        return co.conv_fwd(xs[it[0] % 6], wf, 1, 1, stat_part=part)

    def dg():
        it[0] += 1
        return co.conv_dgrad(xs[it[0] % 6], wd, 1, 1, (h, h), red=(xr[it[0] % 3], pr, coef))
    fl = 2 * M * c * c * 9
    for nm, fn in (("fwd+stats", fwd), ("dgrad+mask+sums", dg)):
        r = {}
        for v in (7, 9):
            lib.ppv_conv_set_variant(v)
            r[v] = timed(fn)
        lib.ppv_conv_set_variant(0)
        print(f"3x3 {c}->{c} @{h}x{h} {nm}: tiled {r[7]:6.1f} us ({fl / r[7] / 1e6:4.0f} TF/s)   halo {r[9]:6.1f} us ({fl / r[9] / 1e6:4.0f} TF/s)")
