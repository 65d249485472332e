"""A/B of the wgrad kernel variants on layer-2/3/4 shapes with cold (rotating) operands (times include the slab reduce)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
B = 128
SH = [(256, 1024, 1, 16), (1024, 256, 1, 16), (256, 256, 3, 16), (128, 512, 1, 32), (512, 128, 1, 32), (128, 128, 3, 32), (512, 512, 3, 8), (512, 2048, 1, 8)]
NB = 5
variants = [int(v, 0) for v in sys.argv[1:]] or [0, 6, 3, 7]
for cin, cout, k, h in SH:
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    gs = [torch.randn(B, h, h, cout, device="cuda").bfloat16() for _ in range(NB)]
    acc = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
    fl = 2.0 * B * h * h * cout * cin * k * k
    by = 2.0 * B * h * h * (cin + cout)
    line = f"cin {cin:5d} cout {cout:5d} k{k} h{h:3d} (HBM floor {by/6e6:5.1f} us):"
    ref = None
    for v in variants:
        if (v & 0xff) == 3 and cout % 256:
            continue
        co.L().ppv_wgrad_set_variant(v)
        for i in range(NB):
            o = co.conv_wgrad(gs[i], xs[i], k, k, 1, (k - 1) // 2, scratch=acc)
        if ref is None:
            ref = co.conv_wgrad(gs[0], xs[0], k, k, 1, (k - 1) // 2, scratch=acc).clone()
            err = 0.0
        else:
            o0 = co.conv_wgrad(gs[0], xs[0], k, k, 1, (k - 1) // 2, scratch=acc)
            err = ((o0 - ref).abs().max() / ref.abs().max()).item()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(4 * NB):
            co.conv_wgrad(gs[i % NB], xs[i % NB], k, k, 1, (k - 1) // 2, scratch=acc)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (4 * NB)
        line += f"  v{v:#x} {ms*1e3:6.1f}us {fl/ms/1e9:4.0f}TF" + (f" (diff {err:.0e})" if err else "")
    print(line)
co.L().ppv_wgrad_set_variant(0)
