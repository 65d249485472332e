"""Per-step time of the kernels the caption decoder adds (rocprofv3 kernel_stats.csv of `bench.py --decoder`).
usage: python tools/decoder_kernels.py <kernel_stats.csv> <steps_recorded>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
S = float(sys.argv[2])
keys = ("dec", "lstm", "gemm_f32", "Cijk", "softmax", "elementwise", "reduce", "index", "cat", "embedding", "nll", "gather", "sort", "copy", "ill", "sigmoid", "tanh", "mul", "add")
tot = 0
for r in rows:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("ppv::", "")
    if any(k in n for k in keys) and "conv" not in n and "bn_" not in n:
        t = float(r["TotalDurationNs"]) / S / 1e3
        tot += t
        if t > 15:
            print(f"{t:8.1f} us/step {int(r['Calls']) / S:7.1f} x {float(r['AverageNs']) / 1e3:7.1f} us  {n[:110]}")
print("sum of listed families: %.1f us/step" % tot)
