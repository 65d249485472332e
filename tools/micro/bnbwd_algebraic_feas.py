import sys, os
sys.path.insert(0, "/root/repo")
import torch
import ppv_amd.convops as co
B, h = 128, 16
M = B * h * h
def timed(fn, nb, reps=3):
    for i in range(nb): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps * nb): fn(i % nb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * nb) * 1e3
co.zero_page(torch.device("cuda", 0))
for (hh, c4, c) in [(16, 1024, 256), (32, 512, 128), (64, 256, 64), (8, 2048, 512)]:
    M = B * hh * hh
    for K in (c4, 2 * c4 + 64):
        NB = max(3, int(800e6 // (M * (K + c) * 2)) + 1)
        xs = [torch.randn(B, hh, hh, K, device="cuda").bfloat16() for _ in range(NB)]
        w = co.weight_layout(torch.randn(c, K, 1, 1, device="cuda") * 0.05, 0)
        xr = torch.randn(B, hh, hh, c, device="cuda").bfloat16()
        coef = torch.rand(4, c, device="cuda")
        part = torch.zeros(64 * c, device="cuda")
        if co.red_supported(M, c):
            t = timed(lambda i: co.conv_dgrad(xs[i], w, 1, 0, (hh, hh), red=(xr, part, coef)), NB)
        else:
            t = timed(lambda i: co.conv_dgrad(xs[i], w, 1, 0, (hh, hh)), NB)
        print(f"h{hh} dgrad3-like K={K} -> N={c}: {t:.1f} us", flush=True)
    # T-sized bn backward apply
    NB = max(3, int(800e6 // (M * c4 * 6)) + 1)
    gs = [torch.randn(B, hh, hh, c4, device="cuda").bfloat16() for _ in range(NB)]
    xs = [torch.randn(B, hh, hh, c4, device="cuda").bfloat16() for _ in range(NB)]
    coef4 = torch.rand(4, c4, device="cuda")
    part = torch.zeros(64 * c4, device="cuda")
    t = timed(lambda i: co.bn_bwd(gs[i], None, xs[i], coef4, 0, want_affine=True, part=part, part_ready=True), NB)
    print(f"h{hh} bn_bwd apply T-size C={c4}: {t:.1f} us", flush=True)
    if c % 128 == 0:
        a2 = [torch.randn(B, hh, hh, c, device="cuda").bfloat16() for _ in range(8)]
        scratch = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
        t = timed(lambda i: co.conv_wgrad(a2[i], a2[i], 1, 1, 1, 0, scratch=scratch), 8)
        print(f"h{hh} Gram(a2) c={c}: {t:.1f} us (incl. slab reduce)", flush=True)
