"""VERDICT r4 task 6, measured with the kernels that exist: would a layer-3 bottleneck gain from never storing x3 = conv3(a2)?
The structural form needs, per block,
  forward : the P x P Gram matrix of a2 (bn3's statistics come from it), then conv3 with BatchNorm + residual (+ ReLU) in its epilogue
            INSTEAD OF conv3 (+ statistics in its epilogue) and the element-wise bn3 + residual + ReLU pass;
  backward: g_a2 = g_m (s o W3) - a2 Q + const INSTEAD OF bn3's backward apply pass + conv3's data gradient on its output
            (Q = W3^T diag(s b / sigma) W3, P x P: the second product is a P -> P 1x1 convolution whose result is the first one's addend).
Lower bounds of the new form from existing launches (cold, rotating operands, layer-3 shape M = 128 x 16 x 16, P = 256):
  Gram        = ppv_conv_wgrad with G = X = a2 (the split-M transposed-read kernel + its slab reduce)
  conv3 + res = ppv_conv_gemm 256 -> 1024 with an addend tile (the epilogue that exists; no scale / shift / ReLU arithmetic yet)
  a2 Q        = ppv_conv_gemm 256 -> 256 (1x1), its result the addend of the 1024 -> 256 data gradient
The coefficient kernels of the new form (w_c^T Cov w_c per output channel; Q) are NOT counted: >= 2 more launches per block.
Writes profiles/r05_gram_bn3_ab.json (or argv[1])."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import ppv_amd.convops as co

B, H, P = 128, 16, 256
M = B * H * H
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_gram_bn3_ab.json")
NB = 6                                                         # 6 x (16.8 + 67 + 67 MB) > the 256-MB Infinity Cache


def timeit(fn, reps=5):
    for i in range(NB):
        fn(i)
    torch.cuda.synchronize()
    meds = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps * NB):
            fn(i % NB)
        e1.record()
        torch.cuda.synchronize()
        meds.append(e0.elapsed_time(e1) / (reps * NB) * 1e3)
    return round(sorted(meds)[1], 1)


a2 = [torch.randn(B, H, H, P, device="cuda").relu().bfloat16() for _ in range(NB)]
res = [torch.randn(B, H, H, 4 * P, device="cuda").bfloat16() for _ in range(NB)]
gm = [torch.randn(B, H, H, 4 * P, device="cuda").bfloat16() for _ in range(NB)]
x3 = [torch.randn(B, H, H, 4 * P, device="cuda").bfloat16() for _ in range(NB)]
w3 = torch.randn(4 * P, P, 1, 1, device="cuda") * 0.05
wt3 = co.weight_layout(w3, 0)                                  # forward layout [Cout][R][S][Cin]
wd3 = co.weight_layout(w3, 1)                                  # data-gradient layout [Cin][R][S][Cout]
q = torch.randn(P, P, 1, 1, device="cuda") * 0.05
wq = co.weight_layout(q, 0)
bn = torch.nn.BatchNorm2d(4 * P).cuda()
sums = torch.zeros(4, 2, 4 * P, device="cuda")               # the step folds the statistics into <= 8 partial rows
part = torch.zeros(64 * 4 * P, device="cuda")
scratch = torch.empty(co.wgrad_scratch_bytes(M, P, 1, 1, P), dtype=torch.uint8, device="cuda")
gram = torch.empty(P, P, 1, 1, device="cuda")
r = {}

# ---- forward, as shipped
r["fwd_old_conv3_with_stats_us"] = timeit(lambda i: co.conv_fwd(a2[i], wt3, 1, 0, stat_part=sums))
x3raw = co.conv_fwd(a2[0], wt3, 1, 0, stat_part=sums)
sums.zero_(); co.conv_fwd(a2[0], wt3, 1, 0, stat_part=sums)
r["fwd_old_bn3_res_relu_pass_us"] = timeit(lambda i: co.bn_act_fold(x3[i], sums, M, bn, 0.1, res=res[i], relu=True, want_bits=True))
# ---- forward, structural form (lower bound)
r["fwd_new_gram_us"] = timeit(lambda i: co.conv_wgrad(a2[i], a2[i], 1, 1, 1, 0, scratch=scratch, out=gram))
# a 1x1 forward conv with an addend tile: the data-gradient entry of the same GEMM (g = a2, flipped layout of W3^T)
wt3_as_dgrad = co.weight_layout(w3.permute(1, 0, 2, 3).contiguous(), 1)     # "conv" P <- 4P whose data gradient is a2 W3^T
r["fwd_new_conv3_with_addend_us"] = timeit(lambda i: co.conv_dgrad(a2[i], wt3_as_dgrad, 1, 0, (H, H), addend=res[i]))
# ---- backward, as shipped: bn3's apply pass (sums already taken by the previous launch) + conv3's data gradient
coef = co.bn_act_fold(x3[0], sums, M, bn, 0.1, res=res[0], relu=True, want_bits=True)[2]
part.zero_()
r["bwd_old_bn3_apply_us"] = timeit(lambda i: co.bn_bwd(gm[i], None, x3[i], coef, False, want_affine=True, part=part, part_ready=True))
r["bwd_old_conv3_dgrad_us"] = timeit(lambda i: co.conv_dgrad(gm[i], wd3, 1, 0, (H, H)))
# ---- backward, structural form: a2 Q (P -> P), then the data gradient with that as its addend
t = co.conv_fwd(a2[0], wq, 1, 0)
r["bwd_new_a2_Q_us"] = timeit(lambda i: co.conv_fwd(a2[i], wq, 1, 0))
r["bwd_new_conv3_dgrad_with_addend_us"] = timeit(lambda i: co.conv_dgrad(gm[i], wd3, 1, 0, (H, H), addend=t))
# the weight-gradient product R = g_m^T a2 now precedes the data gradient (its row dots give sum g x3): it moves from the side stream
# onto the main chain
sc3 = torch.empty(co.wgrad_scratch_bytes(M, 4 * P, 1, 1, P), dtype=torch.uint8, device="cuda")
dw = torch.empty(4 * P, P, 1, 1, device="cuda")
r["bwd_wgrad3_now_on_the_main_chain_us"] = timeit(lambda i: co.conv_wgrad(gm[i], a2[i], 1, 1, 1, 0, scratch=sc3, out=dw))

r["fwd_old_us"] = round(r["fwd_old_conv3_with_stats_us"] + r["fwd_old_bn3_res_relu_pass_us"], 1)
r["fwd_new_lower_bound_us"] = round(r["fwd_new_gram_us"] + r["fwd_new_conv3_with_addend_us"], 1)
r["bwd_old_us"] = round(r["bwd_old_bn3_apply_us"] + r["bwd_old_conv3_dgrad_us"], 1)
r["bwd_new_lower_bound_us"] = round(r["bwd_new_a2_Q_us"] + r["bwd_new_conv3_dgrad_with_addend_us"], 1)
for k, v in r.items():
    print(f"{k:45s} {v}")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump({"what": __doc__.split("\n")[0], "shape": f"layer 3: M = {M}, P = {P}, B = {B}; cold rotating operands; us per launch, median of 3 loops",
           "us": r}, open(out_path, "w"), indent=1)
