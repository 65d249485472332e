"""Condense a rocprofv3 kernel_trace.csv of `bench.py` into the timeline of its LAST step: per kernel class the busy time,
and for the device the union of busy intervals, the idle gaps and the time during which >1 kernel was resident.
usage: python tools/step_timeline.py <kernel_trace.csv> <out.json> [steps_recorded]"""
import csv, json, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
# the step boundary: the Zernike contraction opens every step's camera forward
marks = [i for i, e in enumerate(ev) if "zernike_contract_kernel" in e[2]]
lo = marks[-1]
step = ev[lo:]
t0, t1 = step[0][0], max(e[1] for e in step)
busy = defaultdict(float); n = defaultdict(int)
for s, e, k, q in step:
    k = re.sub(r"\(.*", "", k).replace("void ", "")
    busy[k] += e - s; n[k] += 1
# union / overlap by sweep
pts = sorted([(s, 1) for s, e, k, q in step] + [(e, -1) for s, e, k, q in step])
depth, last, union, multi = 0, t0, 0, 0
gaps, gap_at = [], []
ends = sorted((e, k) for s, e, k, q in step)
starts = sorted((s, k) for s, e, k, q in step)
import bisect
for t, d in pts:
    if depth >= 1: union += t - last
    if depth >= 2: multi += t - last
    if depth == 0 and t > last:
        gaps.append(t - last)
        if t - last > 5000:
            i = bisect.bisect_right(ends, (last, "~")) - 1
            j = bisect.bisect_left(starts, (t, ""))
            short = lambda k: re.sub(r"\(.*", "", k).replace("void ", "").replace("ppv::", "")[:60]
            gap_at.append([round((t - last) / 1e3, 1), round((last - t0) / 1e3), short(ends[i][1]), short(starts[j][1])])
    depth += d; last = t
out = {"step_wall_us": (t1 - t0) / 1e3, "device_busy_us": union / 1e3, "overlapped_us": multi / 1e3, "idle_us": (t1 - t0 - union) / 1e3,
       "launches": len(step), "idle_gaps_over_5us": sum(1 for g in gaps if g > 5000), "idle_in_gaps_over_5us_us": sum(g for g in gaps if g > 5000) / 1e3,
       "queues": sorted({q for *_, q in step}), "gaps_over_5us": gap_at,
       "by_kernel_us": {k: [round(v / 1e3, 1), n[k]] for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:45]}}
# run-length sequence of the step's launches (kernel, queue): where the small launches sit
seq, prev = [], None
for s_, e_, k_, q_ in step:
    k_ = re.sub(r"\(.*", "", k_).replace("void ", "").replace("ppv::", "")[:70]
    if prev is not None and prev[0] == k_ and prev[1] == q_:
        prev[2] += 1
    else:
        prev = [k_, q_, 1]
        seq.append(prev)
out["sequence"] = [f"{n}x q{q} {k}" for k, q, n in seq]
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k not in ("by_kernel_us", "gaps_over_5us", "sequence")}))
