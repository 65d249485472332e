"""Print the kernel sequence of the LAST step of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py, run-length
compressed, with per-name counts of the small torch kernels (fills, copies) and the launch that precedes each: finds who enqueues them.
usage: python tools/step_sequence.py <kernel_trace.csv> [marker-substring = zernike_contract]"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "zernike_contract"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if marker in n and "grad" not in n and "sym_check" not in n]
print("marker launches:", len(starts))
a, b = starts[-2], starts[-1]
seq = rows[a:b]
print("kernels in the step:", len(seq), " device time ms:", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq) / 1e6,
      " span ms:", (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e6)
cnt = Counter()
dur = Counter()
for r in seq:
    n = r["Kernel_Name"][:70]
    cnt[n] += 1
    dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, c in cnt.most_common():
    if "at::native" in n or "rocclr" in n or "Cijk" in n:
        print(f"{c:5d} {dur[n] / 1e3:9.1f} us  {n}")
print("---- context of torch / runtime kernels")
prev = None
ctx = Counter()
for i, r in enumerate(seq):
    n = r["Kernel_Name"]
    if "at::native" in n or "rocclr" in n:
        before = seq[i - 1]["Kernel_Name"][:50] if i else "-"
        after = seq[i + 1]["Kernel_Name"][:50] if i + 1 < len(seq) else "-"
        ctx[(n[:60], before, after)] += 1
for (n, bf, af), c in ctx.most_common(60):
    print(f"{c:4d}  {n}\n        after  {bf}\n        before {af}")
