"""Layer-4 (default; argv[1] = 2 / 3: the other layers' block-0 and body shapes) convolutions (B = 128, 8 x 8 maps; conv1 of block 0 on 16 x 16), cold rotating operands: forward with BatchNorm statistics and
data gradient with the fused BN-backward sums.  Run on the GPU box: python tools/bench_layer4.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co

B = 128


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


SHAPES = {"4": [(512, 512, 3, 8, 1), (512, 512, 3, 16, 2), (2048, 512, 1, 8, 1), (512, 2048, 1, 8, 1), (1024, 512, 1, 16, 1),
                (1024, 2048, 1, 16, 2)],
          "3": [(256, 256, 3, 32, 2), (512, 256, 1, 32, 1), (512, 1024, 1, 32, 2), (256, 1024, 1, 16, 1), (1024, 256, 1, 16, 1)],
          "2": [(128, 128, 3, 64, 2), (256, 128, 1, 64, 1), (256, 512, 1, 64, 2), (128, 512, 1, 32, 1), (512, 128, 1, 32, 1)]}
for cin, cout, k, h, stride in SHAPES[sys.argv[1] if len(sys.argv) > 1 else "4"]:
    NB = max(3, min(24, int(800e6 // (B * h * h * cin * 2))))
    ho = h // stride
    M = B * ho * ho
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    gs = [torch.randn(B, ho, ho, cout, device="cuda").bfloat16() for _ in range(NB)]
    xr = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(3)]
    ws = [torch.randn(cout, cin, k, k, device="cuda") * 0.02 for _ in range(4)]
    wf, wd = [co.weight_layout(w, 0) for w in ws], [co.weight_layout(w, 1) for w in ws]
    part = torch.zeros(co.stat_tiles(M), 2, cout, device="cuda")
    pr = torch.zeros(64 * cin, device="cuda")
    coef = torch.rand(4, cin, device="cuda") + 0.5
    it = [0]
    pad = (k - 1) // 2

    def fwd():
        it[0] += 1
        return co.conv_fwd(xs[it[0] % NB], wf[it[0] % 4], stride, pad, stat_part=part)

    def dg():
        it[0] += 1
        return co.conv_dgrad(gs[it[0] % NB], wd[it[0] % 4], stride, pad, (h, h), red=(xr[it[0] % 3], pr, coef))
    fl = 2 * M * cin * cout * k * k
    for nm, fn in (("fwd+stats", fwd), ("dgrad+mask+sums", dg)):
        t = timed(fn)
        print(f"{k}x{k} {cin}->{cout} @{h}x{h}/{stride} {nm}: {t:6.1f} us  ({fl / t / 1e6:5.0f} TF/s)")
