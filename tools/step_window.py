"""Print a time window of the LAST step of a rocprofv3 kernel_trace.csv (bench.py): start (us from the step's first kernel), duration, queue,
kernel -- to see how the main chain and the weight-gradient stream interleave.  usage: step_window.py trace.csv from_us to_us"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
marks = [i for i, e in enumerate(ev) if "zernike_contract" in e[2]]
step = ev[marks[-1]:]
t0 = step[0][0]
lo, hi = float(sys.argv[2]), float(sys.argv[3])
qs = sorted({q for *_, q in step})
print("step wall us", (max(e[1] for e in step) - t0) / 1e3, "queues", qs)
for s, e, k, q in step:
    a = (s - t0) / 1e3
    if lo <= a <= hi:
        k = re.sub(r"\(.*", "", k).replace("void ", "").replace("ppv::", "")[:64]
        print(f"{a:9.1f} +{(e - s) / 1e3:7.1f}  q{qs.index(q)}  {'    ' * qs.index(q)}{k}")
