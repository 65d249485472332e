"""Fold the passes of tools/pmc_study.sh: per kernel (selected classes) the average counter value per launch."""
import csv, glob, os, re, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("ppv::", "")
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
want = ["conv_wgrad_pipe_kernel<128, 2, false>", "conv_wgrad3x3_kernel<3>", "conv_gemm_pipe_kernel<256, 128, 3, 64, 1, false, false>",
        "conv_gemm_pipe_kernel<256, 128, 3, 32, 2, false, false>", "conv3x3_halo_kernel<false>", "conv1x1_stream_kernel<256, false, true, true, true>",
        "bn_bwd_apply_fused_kernel<0, false, false>", "bn_act_kernel<1, true>"]
for k in want:
    if k not in agg:
        continue
    c = {n: v[0] / max(v[1], 1) for n, v in agg[k].items()}
    n = max(v[1] for v in agg[k].values())
    print(f"== {k}  (launches seen {n})")
    for name in sorted(c):
        print(f"   {name:38s} {c[name]:16.1f}")
    if c.get("TCC_REQ"):
        print(f"   -> L2 hit rate {c.get('TCC_HIT', 0) / max(c.get('TCC_HIT', 0) + c.get('TCC_MISS', 0), 1):.3f}")
    if c.get("TCP_TCC_READ_REQ"):
        print(f"   -> mean TCP->TCC read latency {c.get('TCP_TCC_READ_REQ_LATENCY', 0) / c['TCP_TCC_READ_REQ']:.0f} cycles")
    if c.get("TCP_UTCL1_REQUEST"):
        print(f"   -> UTCL1 miss rate {c.get('TCP_UTCL1_TRANSLATION_MISS', 0) / c['TCP_UTCL1_REQUEST']:.4f}")
