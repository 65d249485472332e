"""Run-length listing of the LAST bench step's launches in start order (rocprofv3 kernel_trace.csv): where the small fills / copies sit.
usage: python tools/step_sequence.py <kernel_trace.csv> <out.txt>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
lo = [i for i, e in enumerate(ev) if "zernike_contract_kernel" in e[2]][-1]
step = ev[lo:]
t0 = step[0][0]
short = lambda k: re.sub(r"\(.*", "", k).replace("void ", "").replace("ppv::", "").replace("at::native::", "")[:70]
out, prev, n, tstart, dur = [], None, 0, 0, 0
for s, e, k, q in step:
    k = short(k) + " q" + q
    if k == prev:
        n += 1; dur += e - s
    else:
        if prev: out.append(f"{(tstart - t0) / 1e3:9.1f} us  x{n:<3d} {dur / 1e3:8.1f} us  {prev}")
        prev, n, tstart, dur = k, 1, s, e - s
out.append(f"{(tstart - t0) / 1e3:9.1f} us  x{n:<3d} {dur / 1e3:8.1f} us  {prev}")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print(len(out), "runs")
