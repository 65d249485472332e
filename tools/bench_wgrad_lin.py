"""A/B of the linear-address 1x1 weight-gradient kernel (conv_wgrad_lin_kernel, round 6) against the general kernel it replaces
(conv_wgrad_pipe_kernel<256, 3>, variant bit 0x4000), one process, interleaved, cold rotating operands, B = 128; us per call including
the slab reduce.  Writes profiles/r06_wgrad_lin_ab.json (or argv[1])."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.convops as co

B, NB = 128, 5
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_wgrad_lin_ab.json")
res = {}
for cin, cout, h in [(256, 1024, 16), (1024, 256, 16), (128, 512, 32), (512, 128, 32), (512, 2048, 8), (2048, 512, 8), (1024, 512, 16), (1024, 2048, 8)]:
    if cout % 256:
        continue
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(NB)]
    gs = [torch.randn(B, h, h, cout, device="cuda").bfloat16() for _ in range(NB)]
    scratch = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
    dw = torch.empty(cout, cin, 1, 1, device="cuda")

    def t(variant, n=6 * NB):
        co.L().ppv_wgrad_set_variant(variant)
        for i in range(NB):
            co.conv_wgrad(gs[i], xs[i], 1, 1, 1, 0, scratch=scratch, out=dw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            co.conv_wgrad(gs[i % NB], xs[i % NB], 1, 1, 1, 0, scratch=scratch, out=dw)
        e1.record()
        torch.cuda.synchronize()
        co.L().ppv_wgrad_set_variant(0)
        return e0.elapsed_time(e1) / n * 1e3

    lin, gen, pp = [], [], []
    for _ in range(3):
        lin.append(t(0)); gen.append(t(0x4000)); pp.append(t(0x8000))
    key = f"{cin}_to_{cout}_at_{h}"
    res[key] = {"linear_us": round(sorted(lin)[1], 1), "general_us": round(sorted(gen)[1], 1), "other_phase_schedule_us": round(sorted(pp)[1], 1)}
    print(key, res[key], flush=True)
json.dump({"what": __doc__.split("\n\n")[0].replace("\n", " "), "us_per_call_incl_reduce": res}, open(out, "w"), indent=1)
