"""Cold-operand A/B of the conv GEMM variants on the shapes that carry the step (weights = launches per step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
B = 128
SH = [(256, 1024, 1, 16, 46), (1024, 256, 1, 16, 45), (256, 256, 3, 16, 45), (128, 512, 1, 32, 8), (512, 128, 1, 32, 8),
      (128, 128, 3, 32, 8), (512, 2048, 1, 8, 6), (2048, 512, 1, 8, 5), (512, 512, 3, 8, 5), (64, 256, 1, 64, 6), (256, 64, 1, 64, 5)]
variants = [int(v) for v in sys.argv[1:]] or [3, 4, 5, 7]
tot = {v: 0.0 for v in variants}
for cin, cout, k, h, wgt in SH:
    NB = max(2, int(600e6 // (B * h * h * (cin + cout) * 2)) + 1)
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    w = co.weight_layout(torch.randn(cout, cin, k, k, device="cuda") * 0.05, 0)
    part = torch.zeros(co.stat_tiles(B * h * h), 2, cout, device="cuda")
    fl = 2.0 * B * h * h * cout * cin * k * k
    line = f"cin {cin:5d} cout {cout:5d} k{k} h{h:3d} x{wgt:2d}:"
    for v in variants:
        co.L().ppv_conv_set_variant(v)
        for i in range(NB):
            co.conv_fwd(xs[i], w, 1, (k - 1) // 2, stat_part=part)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 3 * NB
        e0.record()
        for i in range(n):
            y = co.conv_fwd(xs[i % NB], w, 1, (k - 1) // 2, stat_part=part)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        tot[v] += ms * wgt
        # bytes a launch stages through LDS (A re-read per column tile, W per row tile) for the tile the auto rule picks, and HBM-compulsory bytes
        M_, bm, bn = B * h * h, (128 if v == 2 else 256), (64 if cout % 128 else 128)
        staged = 2.0 * M_ * cout * cin * k * k * (1.0 / bn + 1.0 / bm)
        hbm = 2.0 * (M_ * cin + M_ * cout + cout * cin * k * k)
        line += f"  v{v} {ms*1e3:6.1f}us {fl/ms/1e9:4.0f}TF staged {staged/ms/1e9:5.2f}TB/s hbm {hbm/ms/1e9:5.2f}TB/s"
    print(line)
co.L().ppv_conv_set_variant(0)
print("weighted ms/step:", {v: round(t, 3) for v, t in tot.items()})
