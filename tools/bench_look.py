"""Round 6: the look-ahead K loop (variant 6: <256,128,6,32,1> with the fragments of step t + 1 read under the MFMAs of step t) against the
plain one-round tile (variant 3: <256,128,3,64,1>) and the two-per-CU tile (variant 4) on the trunk's 1x1 shapes at B = 128, forward with
statistics and data gradient with the fused BN-backward sums; cold rotating operands.  GPU box: python tools/bench_look.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co

B, NB = 128, 6
lib = co.L()


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
for cin, cout, h in [(1024, 256, 16), (256, 1024, 16), (512, 128, 32), (128, 512, 32), (2048, 512, 8), (512, 2048, 8), (1024, 512, 16), (1024, 2048, 8)]:
    M = B * h * h
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    gs = [torch.randn(B, h, h, cout, device="cuda").bfloat16() for _ in range(NB)]
    xr = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(3)]
    w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.03
    wf, wd = co.weight_layout(w, 0), co.weight_layout(w, 1)
    part = torch.zeros(co.stat_tiles(M), 2, cout, device="cuda")
    pr = torch.zeros(64 * cin, device="cuda")
    coef = torch.rand(4, cin, device="cuda") + 0.5
    it = [0]

    def fwd():
        it[0] += 1
        return co.conv_fwd(xs[it[0] % NB], wf, 1, 0, stat_part=part)

    def dg():
        it[0] += 1
        return co.conv_dgrad(gs[it[0] % NB], wd, 1, 0, (h, h), red=(xr[it[0] % 3], pr, coef))

    row = {}
    for nm, fn in (("fwd+stats", fwd), ("dgrad+sums", dg)):
        r = {}
        for v in (0, 3, 4, 6, 11, 12):
            lib.ppv_conv_set_variant(v)
            try:
                r[f"v{v}"] = timed(fn)
            except Exception as e:  # noqa: BLE001
                r[f"v{v}"] = None
        lib.ppv_conv_set_variant(0)
        row[nm] = r
    out[f"{cin}_to_{cout}_at_{h}"] = row
    print(cin, cout, h, row, flush=True)
if len(sys.argv) > 1:
    json.dump({"what": "1x1 conv tiles at B = 128, us per launch (cold operands, median of 3 loops): v0 = automatic rule, v3 = <256,128,3,64,1>, "
                       "v4 = <256,128,3,32,2>, v6 = <256,128,6,32,1> look-ahead, v11 = <256,128,3,64,1> ping-pong (round 6)", "us": out}, open(sys.argv[1], "w"), indent=1)
