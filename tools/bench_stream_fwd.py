import sys, os
sys.path.insert(0, "/root/repo")
import torch
import ppv_amd.convops as co
B=128
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cin, cout, h in [(256, 1024, 16), (128, 512, 32), (512, 2048, 8), (64, 256, 64)]:
    M = B*h*h
    NB = 12
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    outs = [torch.empty(B, h, h, cout, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
    w = co.weight_layout(torch.randn(cout, cin, 1, 1, device="cuda") * 0.02, 0)
    part = torch.zeros(co.stat_tiles(M), 2, cout, device="cuda")
    it = [0]
    def fwd():
        it[0] += 1
        return co.conv_fwd(xs[it[0] % NB], w, 1, 0, stat_part=part)
    r = {}
    for v in (0, 8):
        co.L().ppv_conv_set_variant(v)
        r[v] = timed(fwd)
    co.L().ppv_conv_set_variant(0)
    print(f"1x1 {cin}->{cout} @{h} fwd+stats: auto {r[0]:.1f} us  stream {r[8]:.1f} us")
