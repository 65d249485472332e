"""Round 6 A/B: bn1 + ReLU inside conv2's halo kernel (ppv_conv3x3_bnin) against the two launches it replaces (ppv_bn_act_fold_rows +
ppv_conv_gemm on the halo kernel), layer-3 / layer-2 shapes at B = 128, cold rotating operands.  GPU box: python tools/bench_bnin.py [out.json]"""
import copy
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co

B, NB = 128, 8


def timed(fn, n=24):
    for _ in range(NB):
        fn()
    torch.cuda.synchronize()
    meds = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        meds.append(e0.elapsed_time(e1) / n * 1e3)
    return round(sorted(meds)[1], 1)


out = {}
for c, h in [(256, 16), (128, 32)]:
    xs = [torch.randn(B, h, h, c, device="cuda").bfloat16() for _ in range(NB)]
    w = torch.randn(c, c, 3, 3, device="cuda") * 0.02
    wf = co.weight_layout(w, 0)
    M = B * h * h
    bn = torch.nn.BatchNorm2d(c).cuda()
    xf = xs[0].float().view(-1, c)
    sums = torch.stack([torch.stack([xf.sum(0), (xf ** 2).sum(0)]), torch.zeros(2, c, device="cuda")]).contiguous()
    part = torch.zeros(co.stat_tiles(M), 2, c, device="cuda")
    it = [0]

    def two():
        it[0] += 1
        y, _, _ = co.bn_act_fold(xs[it[0] % NB], sums, M, bn, 0.1)
        return co.conv_fwd(y, wf, 1, 1, stat_part=part)

    def conv_only():
        it[0] += 1
        return co.conv_fwd(xs[it[0] % NB], wf, 1, 1, stat_part=part)

    def fold_only():
        it[0] += 1
        return co.bn_act_fold(xs[it[0] % NB], sums, M, bn, 0.1)

    def one():
        it[0] += 1
        return co.conv3x3_bnin(xs[it[0] % NB], sums, M, bn, 0.1, wf, stat_part=part)

    def one_noact():
        it[0] += 1
        return co.conv3x3_bnin(xs[it[0] % NB], sums, M, bn, 0.1, wf, stat_part=part, want_act=False)

    r = {"bn_apply_plus_conv_us": timed(two), "conv_alone_us": timed(conv_only), "bn_apply_alone_us": timed(fold_only), "bnin_us": timed(one),
         "bnin_without_y_store_us": timed(one_noact)}
    out[f"3x3_{c}_{h}x{h}"] = r
    print(c, h, r, flush=True)
if len(sys.argv) > 1:
    json.dump({"what": "bn1 + ReLU in conv2's LDS halo tile (ppv_conv3x3_bnin) vs ppv_bn_act_fold_rows + ppv_conv_gemm, B = 128, cold operands, us per call "
                       "(median of 3 loops of 24; includes allocator + launch gaps of back-to-back calls)", "us": out}, open(sys.argv[1], "w"), indent=1)
