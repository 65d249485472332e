"""A/B of the wgrad kernel variants on layer-2/3/4 shapes with cold (rotating) operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co
B = 128
SH = [(256, 1024, 1, 16), (1024, 256, 1, 16), (256, 256, 3, 16), (128, 512, 1, 32), (128, 128, 3, 32), (512, 512, 3, 8), (512, 2048, 1, 8)]
NB = 5
for cin, cout, k, h in SH:
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    gs = [torch.randn(B, h, h, cout, device="cuda").bfloat16() for _ in range(NB)]
    acc = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    fl = 2.0 * B * h * h * cout * cin * k * k
    line = f"cin {cin:5d} cout {cout:5d} k{k} h{h:3d}:"
    for v in (0, 1, 3, 0x103):
        if v and (v & 0xff) == 3 and cout % 256:
            continue
        co.L().ppv_wgrad_set_variant(v)
        for i in range(NB):
            co.conv_wgrad(gs[i], xs[i], k, k, 1, (k - 1) // 2, scratch=acc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(2 * NB):
            co.conv_wgrad(gs[i % NB], xs[i % NB], k, k, 1, (k - 1) // 2, scratch=acc)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (2 * NB)
        line += f"  v{v:#x} {ms*1e3:6.1f}us {fl/ms/1e9:4.0f}TF"
    print(line)
co.L().ppv_wgrad_set_variant(0)
