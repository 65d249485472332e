"""A/B of the cooperative (grid-barrier) conv + BatchNorm + ReLU launch against the two-launch form it would replace (VERDICT r3 task 1b).

Shape: conv1 of a layer-3 identity bottleneck at the benchmark size -- [128,16,16,1024] x [256,1024] -> raw x1 and y1 = relu(bn1(x1)),
both bf16, train-mode statistics (Image_Caption/models.py:17-21, train.py:245).
  A (product): ppv_conv_gemm with two partial statistic rows  +  ppv_bn_act_fold_rows            (two launches)
  B (coop)   : ppv_conv_bn_relu_coop: same tile, statistics -> grid barrier -> apply on the tile in LDS, both tensors stored (one launch)
Cold operands: NB input / output sets rotate (more bytes than the Infinity Cache), events around n launches, both forms interleaved.
Prints one JSON line (profiles/r04*_coop_ab.json)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd  # noqa: F401
import ppv_amd.convops as co
from ppv_amd import _lib
from ppv_amd._lib import check, ptr, stream_ptr

dev = torch.device("cuda", 0)
B, H, CIN, COUT = 128, 16, 1024, 256
M = B * H * H
NB = 10
g = torch.Generator().manual_seed(0)
xs = [torch.randn(B, H, H, CIN, generator=g).bfloat16().to(dev) for _ in range(NB)]
w = co.weight_layout((torch.randn(COUT, CIN, 1, 1, generator=g) * 0.03).to(dev), 0)
bn = torch.nn.BatchNorm2d(COUT).to(dev).train()
with torch.no_grad():
    bn.weight.uniform_(0.5, 1.5)
    bn.bias.uniform_(-0.2, 0.2)
ROWS = 2
L = _lib.lib()
zp = co.zero_page(dev)
x1 = [torch.empty(B, H, H, COUT, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
y1 = [torch.empty(B, H, H, COUT, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
coef = torch.empty(4, COUT, dtype=torch.float32, device=dev)
NL = 400                                                   # launches per form: every launch its own pre-zeroed statistics / counter
stats = torch.zeros(NL, ROWS, 2, COUT, dtype=torch.float32, device=dev)
counters = torch.zeros(NL, 2, dtype=torch.int32, device=dev)
rm, rv = bn.running_mean, bn.running_var


def form_a(i, k):
    check(L.ppv_conv_gemm(ptr(xs[i]), ptr(w), ptr(x1[i]), ptr(stats[k]), None, None, ptr(zp), B, H, H, CIN, H, H, COUT, 1, 1, 1, 0, 1, 0, ROWS,
                          stream_ptr()), "conv")
    check(L.ppv_bn_act_fold_rows(ptr(x1[i]), ptr(stats[k]), ROWS, float(M), ptr(bn.weight), ptr(bn.bias), ptr(rm), ptr(rv), 0.1, bn.eps, ptr(coef),
                                 None, ptr(y1[i]), None, M * COUT, COUT, 0, 1, stream_ptr()), "bn_act_fold")


def form_b(i, k):
    check(L.ppv_conv_bn_relu_coop(ptr(xs[i]), ptr(w), ptr(x1[i]), ptr(y1[i]), ptr(stats[k]), ptr(counters[k]), ptr(bn.weight), ptr(bn.bias), ptr(rm), ptr(rv),
                                  0.1, bn.eps, ptr(coef), ptr(zp), B, H, H, CIN, COUT, ROWS, stream_ptr()), "coop")


# ---- parity of the two forms on the same input
stats.zero_(); counters.zero_()
form_a(0, 0)
torch.cuda.synchronize()
ya, xa, ca = y1[0].clone(), x1[0].clone(), coef.clone()
form_b(0, 1)
torch.cuda.synchronize()
yb, xb, cb = y1[0].clone(), x1[0].clone(), coef.clone()
timed_out = int(counters[1, 1].item())
raw_equal = bool(torch.equal(xa, xb))
y_err = float((ya.float() - yb.float()).abs().max() / ya.float().abs().max())
coef_err = float((ca - cb).abs().max() / ca.abs().max())
assert timed_out == 0, "the grid barrier did not complete"
assert raw_equal and y_err < 2 ** -7 and coef_err < 1e-4, (raw_equal, y_err, coef_err)


def timeit(fn, n):
    stats.zero_(); counters.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n):
        fn(k % NB, k)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {"a": [], "b": []}
for rep in range(5):                                       # interleaved: a drifting box shows in both
    res["a"].append(timeit(form_a, NL))
    res["b"].append(timeit(form_b, NL))
bad = int(counters[:, 1].sum().item())
med = lambda v: sorted(v)[len(v) // 2]
a, b = med(res["a"]), med(res["b"])
print(json.dumps({
    "shape": f"conv1 of a layer-3 identity bottleneck: [{B},{H},{H},{CIN}] x [{COUT},{CIN}] + BatchNorm(train) + ReLU, bf16, cold operands ({NB} rotating sets)",
    "two_launches_us": round(a, 2), "coop_one_launch_us": round(b, 2), "gain_us": round(a - b, 2), "gain_frac_of_pair": round((a - b) / a, 4),
    "gain_frac_of_block_forward_180us": round((a - b) / 180.0, 4),
    "runs_us": {k: [round(x, 2) for x in v] for k, v in res.items()},
    "parity": {"raw_bitwise_equal": raw_equal, "y_max_rel_err": y_err, "coef_max_rel_err": coef_err},
    "barrier_timeouts": bad,
    "adopt_bar": "8 % of the block (VERDICT r3 task 1b)"}))
