"""Group the launches of the LAST step of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py by (kernel, grid size): in-step
duration of every launch shape.  usage: python tools/step_by_shape.py <kernel_trace.csv> [min total us = 100]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "zernike_contract" in r["Kernel_Name"]]
pairs = [(a, b) for a, b in zip(marks, marks[1:]) if b - a > 500]        # a whole training step (the camera-only legs that follow are short)
seq = rows[pairs[-1][0]:pairs[-1][1]]
agg = defaultdict(list)
for r in seq:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("ppv::", "")
    agg[(n[:64], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"launches {len(seq)}  device time {tot / 1e3:.3f} ms")
for (n, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) >= floor:
        print(f"{len(v):4d} x {sum(v) / len(v):7.1f} us = {sum(v):8.1f} us  [{min(v):6.1f} .. {max(v):6.1f}]  grid {grid:6d}  {n}")
