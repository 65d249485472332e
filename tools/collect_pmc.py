"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of
`bench.py --steps 1 --warmup 1` into profiles/<name>.json: HBM-side bytes per launch for every kernel.

usage: collect_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads, guide)."""
import csv, json, os, re, sys, collections

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_hash


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            a = agg[name]
            a[0] += float(row["Counter_Value"]) * 1024.0
            a[1] += 1
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on `bench.py --steps 1 --warmup 1` (2 steps recorded); "
               "counter unit KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)",
       "csrc_sha16": csrc_hash(), "per_kernel": {}}
tot_f = tot_w = 0.0
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch[k][0] + write[k][0])):
    n = max(fetch[k][1], write[k][1], 1)
    out["per_kernel"][k] = {"launches_2steps": n, "fetch_bytes_per_launch_corrected": 2 * fetch[k][0] / n,
                            "write_bytes_per_launch": write[k][0] / n}
    tot_f += 2 * fetch[k][0]
    tot_w += write[k][0]
out["total_fetch_GB_per_step"] = round(tot_f / 2 / 1e9, 2)
out["total_write_GB_per_step"] = round(tot_w / 2 / 1e9, 2)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("fetch GB/step", out["total_fetch_GB_per_step"], "write GB/step", out["total_write_GB_per_step"])
