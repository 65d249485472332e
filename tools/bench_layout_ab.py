"""VERDICT r2 task 5: prove or kill the channel-chunked activation layout on one shape before touching the trunk.

Standalone A/B, cold (rotating) operands, one process, interleaved rounds: the layer-3 conv1 / conv3 forward launches on the
256 x 128 BK-64 tile (ppv_conv_set_variant 3) and the 1x1 weight gradients, with their activation operands stored NHWC ([M][C]) against
channel-chunked ([C/64][M][64] for the convolution's A operand: a K-step's 256 rows are one contiguous 32-KB block; [C/128][M][128]
for both weight-gradient operands: a 64-row stage is one contiguous 16-KB block per operand).  Outputs must be identical.
Writes a JSON (argv[1], default gpurun_out/layout_ab.json) with both timings per shape and the class-level ratio."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co

B, h = 128, 16
M = B * h * h
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/layout_ab.json"
res = {"note": "layer-3 shapes at B = 128 (M = 32768), cold rotating operands, median of interleaved rounds; times in us", "conv_fwd": {}, "wgrad_1x1": {}}


def chunk(x, cw):                    # [B,h,h,C] -> [C/cw][M][cw] stored behind the same shape
    C = x.shape[-1]
    return x.view(M, C // cw, cw).permute(1, 0, 2).contiguous().view(x.shape)


def timed(fn, nb, rounds=7, reps=3):
    for i in range(nb):
        fn(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps * nb):
            fn(i % nb)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / (reps * nb) * 1e3)
    return ts


co.zero_page(torch.device("cuda", 0))
for cin, cout in [(1024, 256), (256, 1024)]:
    NB = max(3, int(700e6 // (M * (cin + cout) * 2)) + 1)
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    xc = [chunk(x, 64) for x in xs]
    w = co.weight_layout(torch.randn(cout, cin, 1, 1, device="cuda") * 0.05, 0)
    part = torch.zeros(co.stat_tiles(M), 2, cout, device="cuda")

    def run(v, ops):
        def f(i):
            co.L().ppv_conv_set_variant(v)
            return co.conv_fwd(ops[i], w, 1, 0, stat_part=part)
        return f
    a = run(3, xs)(0).clone(); b = run(0x1003, xc)(0).clone()
    assert torch.equal(a, b), "chunked source gives a different result"
    t = {"nhwc_v3": [], "chunked_v3": [], "nhwc_auto": []}
    for _ in range(3):                                   # interleaved
        t["nhwc_v3"] += timed(run(3, xs), NB, rounds=3)
        t["chunked_v3"] += timed(run(0x1003, xc), NB, rounds=3)
        t["nhwc_auto"] += timed(run(0, xs), NB, rounds=3)
    med = {k: round(sorted(v)[len(v) // 2], 2) for k, v in t.items()}
    res["conv_fwd"][f"{cin}->{cout}"] = med
    print("conv fwd", cin, cout, med, flush=True)
co.L().ppv_conv_set_variant(0)

for cin, cout in [(256, 1024), (1024, 256)]:
    NB = max(3, int(700e6 // (M * (cin + cout) * 2)) + 1)
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    gs = [torch.randn(B, h, h, cout, device="cuda").bfloat16() for _ in range(NB)]
    xc, gc = [chunk(x, 128) for x in xs], [chunk(g_, 128) for g_ in gs]
    scratch = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")

    def run(v, go, xo):
        def f(i):
            co.L().ppv_wgrad_set_variant(v)
            return co.conv_wgrad(go[i], xo[i], 1, 1, 1, 0, scratch=scratch)
        return f
    a = run(0, gs, xs)(0).clone(); b = run(0x1000, gc, xc)(0).clone()
    err = ((a - b).abs().max() / a.abs().max()).item()
    assert err < 1e-6, err
    t = {"nhwc": [], "chunked": []}
    for _ in range(3):
        t["nhwc"] += timed(run(0, gs, xs), NB, rounds=3)
        t["chunked"] += timed(run(0x1000, gc, xc), NB, rounds=3)
    med = {k: round(sorted(v)[len(v) // 2], 2) for k, v in t.items()}
    res["wgrad_1x1"][f"{cin}->{cout}"] = med
    print("wgrad", cin, cout, med, flush=True)
co.L().ppv_wgrad_set_variant(0)

cv = res["conv_fwd"]
res["conv_class_ratio_chunked_over_nhwc_v3"] = round(sum(v["chunked_v3"] for v in cv.values()) / sum(v["nhwc_v3"] for v in cv.values()), 3)
res["conv_class_ratio_chunked_over_nhwc_auto"] = round(sum(v["chunked_v3"] for v in cv.values()) / sum(v["nhwc_auto"] for v in cv.values()), 3)
wv = res["wgrad_1x1"]
res["wgrad_ratio_chunked_over_nhwc"] = round(sum(v["chunked"] for v in wv.values()) / sum(v["nhwc"] for v in wv.values()), 3)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if "ratio" in k}))
