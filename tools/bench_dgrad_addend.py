"""Cold A/B of the conv1 data-gradient launches (residual addend + ReLU mask in the store) per tile variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
B = 128
SH = [(1024, 256, 16, 23), (512, 128, 32, 4), (256, 64, 64, 3), (2048, 512, 8, 3)]   # conv1: Cin -> Cout at h (count per step)
variants = [int(v) for v in sys.argv[1:]] or [0]
for cin, cout, h, cnt in SH:
    NB = max(2, int(600e6 // (B * h * h * (3 * cin + cout) * 2)) + 1)
    gs = [torch.randn(B, h, h, cout, device="cuda").bfloat16() for _ in range(NB)]
    adds = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    acts = [torch.randint(0, 256, (B * h * h * cin // 8,), device="cuda", dtype=torch.uint8) for _ in range(NB)]   # (x > 0) bit masks
    xrs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    part = torch.zeros(64 * cin, device="cuda")
    wd = co.weight_layout(torch.randn(cout, cin, 1, 1, device="cuda") * 0.05, 1)
    line = f"dgrad1 {cout:5d} -> {cin:5d} h{h:3d} x{cnt:2d}:"
    for v in variants:
        co.L().ppv_conv_set_variant(v)
        for mode in ("plain", "add+mask", "add+mask+bnred"):
            def run(i):
                if mode == "plain":
                    return co.conv_dgrad(gs[i], wd, 1, 0, (h, h))
                if mode == "add+mask+bnred":                    # also takes the following BN backward's sums against xr
                    return co.conv_dgrad(gs[i], wd, 1, 0, (h, h), addend=adds[i], relu_bits=acts[i], red=(xrs[i], part))
                return co.conv_dgrad(gs[i], wd, 1, 0, (h, h), addend=adds[i], relu_bits=acts[i])
            for i in range(NB):
                run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 3 * NB
            e0.record()
            for i in range(n):
                run(i % NB)
            e1.record(); torch.cuda.synchronize()
            line += f"  v{v} {mode} {e0.elapsed_time(e1) / n * 1e3:6.1f}us"
    print(line)
co.L().ppv_conv_set_variant(0)
