"""Reference ceiling for the forward / data-gradient convolution class (cdna_hip_programming.md 5.4 rule 10: a ceiling claim needs a
known-good reference on the same hardware; VERDICT r4 task 1a).  The 1x1 convolutions of the trunk are plain GEMMs out[M, N] =
x[M, K] w[N, K]^T with M = B*H*W: time hipBLASLt (torch.mm, bf16 in / f32 accumulate / bf16 out) on the benchmark's shapes with COLD
(rotating) operands next to ppv_conv_gemm, and MIOpen (F.conv2d, channels_last bf16) on the two 3x3 shapes that carry the step.
Measurement only -- the product never calls a library here.  Writes profiles/r05_fwd_vs_lib.json (or argv[1])."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import ppv_amd.convops as co

B = 128
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_fwd_vs_lib.json")


def timeit(fn, nb, reps=4):
    for i in range(nb):
        fn(i)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = reps * nb
        e0.record()
        for i in range(n):
            fn(i % nb)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(best)[1]


rows = []
# (K = Cin, N = Cout, map side): layer 3, layer 2, layer 1, layer 4 -- both directions of the bottleneck's two 1x1 convolutions
for cin, cout, h in [(1024, 256, 16), (256, 1024, 16), (512, 128, 32), (128, 512, 32), (256, 64, 64), (64, 256, 64), (2048, 512, 8), (512, 2048, 8)]:
    M = B * h * h
    nb = max(3, int(700e6 // (M * (cin + cout) * 2)) + 1)          # rotate through > 256 MiB (Infinity Cache) of operands
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(nb)]
    w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
    wt = co.weight_layout(w, 0)
    w2 = wt.view(cout, cin)
    part = torch.zeros(co.stat_tiles(M), 2, cout, device="cuda")
    outs = [torch.empty(M, cout, device="cuda", dtype=torch.bfloat16) for _ in range(nb)]
    ours_stats = timeit(lambda i: co.conv_fwd(xs[i], wt, 1, 0, stat_part=part), nb)
    ours_plain = timeit(lambda i: co.conv_fwd(xs[i], wt, 1, 0), nb)
    lib = timeit(lambda i: torch.mm(xs[i].view(M, cin), w2.t(), out=outs[i]), nb)
    fl = 2.0 * M * cin * cout
    by = 2.0 * (M * cin + M * cout + cin * cout)
    r = {"shape": f"1x1 {cin}->{cout} @{h}x{h}", "M": M, "K": cin, "N": cout,
         "ppv_conv_gemm_with_bn_stats_us": round(ours_stats, 1), "ppv_conv_gemm_plain_us": round(ours_plain, 1), "hipblaslt_torch_mm_us": round(lib, 1),
         "lib_over_ours": round(lib / ours_stats, 3), "ours_TFLOPs": round(fl / ours_stats / 1e6, 0), "lib_TFLOPs": round(fl / lib / 1e6, 0),
         "ours_compulsory_TBps": round(by / ours_stats / 1e6, 2), "lib_compulsory_TBps": round(by / lib / 1e6, 2)}
    rows.append(r)
    print(json.dumps(r), flush=True)
    del xs, outs

for c, h in [(256, 16), (128, 32)]:
    M = B * h * h
    nb = max(3, int(700e6 // (M * 2 * c * 2)) + 1)
    xs = [torch.randn(B, h, h, c, device="cuda").bfloat16() for _ in range(nb)]
    w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
    wt = co.weight_layout(w, 0)
    part = torch.zeros(co.stat_tiles(M), 2, c, device="cuda")
    ours = timeit(lambda i: co.conv_fwd(xs[i], wt, 1, 1, stat_part=part), nb)
    xcl = [x.permute(0, 3, 1, 2) for x in xs]                      # NCHW views of NHWC storage = channels_last
    wcl = w.bfloat16().contiguous(memory_format=torch.channels_last)
    try:
        lib = timeit(lambda i: F.conv2d(xcl[i], wcl, padding=1), nb)
    except Exception as e:  # noqa: BLE001
        print("MIOpen conv2d failed:", e)
        lib = float("nan")
    fl = 2.0 * M * c * c * 9
    r = {"shape": f"3x3 {c}->{c} @{h}x{h}", "M": M, "K": 9 * c, "N": c, "ppv_conv_gemm_with_bn_stats_us": round(ours, 1),
         "miopen_conv2d_us": round(lib, 1), "lib_over_ours": round(lib / ours, 3), "ours_TFLOPs": round(fl / ours / 1e6, 0),
         "lib_TFLOPs": round(fl / lib / 1e6, 0)}
    rows.append(r)
    print(json.dumps(r), flush=True)
    del xs, xcl

os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump({"what": "cold-operand us per launch, B = 128; median of 3 timed loops; library = measurement only", "rows": rows}, open(out_path, "w"), indent=1)
