"""Every conv launch of a (1,1,1,1) trunk at B = 2, 64 x 64: repeated launches on the same operands must be bit-identical
(statistics included: at these sizes every row tile owns its partial row)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co

torch.manual_seed(0)
B = 2
shapes = [(16, 64, 64, 1, 1), (16, 64, 64, 3, 1), (16, 64, 256, 1, 1), (16, 256, 128, 1, 1), (16, 128, 128, 3, 2), (8, 128, 512, 1, 1),
          (16, 256, 512, 1, 2), (8, 512, 256, 1, 1), (8, 256, 256, 3, 2), (4, 256, 1024, 1, 1), (8, 512, 1024, 1, 2), (4, 1024, 512, 1, 1),
          (4, 512, 512, 3, 2), (2, 512, 2048, 1, 1), (4, 1024, 2048, 1, 2)]
bad = 0
for h, cin, cout, k, st in shapes:
    x = torch.randn(B, h, h, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    pad = (k - 1) // 2
    ho = (h + 2 * pad - k) // st + 1
    ref = refp = None
    nbad = 0
    for it in range(30):
        wt = co.weight_layout(w, 0)
        part = torch.zeros(co.stat_tiles(B * ho * ho), 2, cout, device="cuda")
        y = co.conv_fwd(x, wt, st, pad, stat_part=part)
        if ref is None:
            ref, refp, refw = y.clone(), part.clone(), wt.clone()
        else:
            if not torch.equal(wt, refw): nbad += 1; print("   weight layout differs")
            if not torch.equal(y, ref):
                nbad += 1
                d = (y.float() - ref.float()).abs()
                print(f"   output differs at iteration {it}: {int((d > 0).sum())} elements, max {d.max().item():.3e}, rows {torch.nonzero(d.sum(-1).flatten() > 0).flatten()[:8].tolist()}")
            if not torch.equal(part, refp): nbad += 1; print("   statistics differ")
    print(f"h {h} {cin}->{cout} k{k} s{st}: {'OK' if not nbad else str(nbad) + ' mismatches'}")
    bad += nbad
print("TOTAL mismatches", bad)
