"""MFMA utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES (rocprofv3 --pmc, its own pass; counts cycles summed over the chip's
1024 SIMDs, MI355X_MICROARCH.md) / (1024 x kernel duration x 2.4 GHz), durations from the --kernel-trace --stats pass.

usage: collect_mfma.py <mfma_counter_collection.csv> <kernel_stats.csv> <out.json>"""
import csv, json, os, re, sys, collections

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_hash

busy = collections.defaultdict(lambda: [0.0, 0])
sq = collections.defaultdict(float)                # SQ_BUSY_CYCLES: summed over the 32 shader engines
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if r.get("Counter_Name") == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[k][0] += float(r["Counter_Value"]); busy[k][1] += 1
    elif r.get("Counter_Name") == "SQ_BUSY_CYCLES":
        sq[k] += float(r["Counter_Value"])
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[re.sub(r"\(.*", "", r["Name"]).replace("void ", "")] = float(r["AverageNs"])
out = {"note": "MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES per launch / (1024 SIMDs x average kernel duration x 2.4 GHz); PMC pass and "
               "kernel-trace pass are separate runs of bench.py (B = 128); *_at_actual_clock = MFMA busy / (32 x SQ_BUSY_CYCLES), i.e. against the cycles the launch really had (the shader clock sits near 1.7-1.9 GHz under MFMA load, not 2.4)", "csrc_sha16": csrc_hash(), "per_kernel": {}}
for k, (c, n) in sorted(busy.items(), key=lambda kv: -kv[1][0]):
    if c <= 0 or k not in dur:
        continue
    out["per_kernel"][k] = {"launches": n, "mfma_busy_cycles_per_launch": c / n, "avg_duration_us": round(dur[k] / 1e3, 2),
                            "mfma_busy_frac": round(c / n / (1024 * dur[k] * 2.4), 4)}
    if sq.get(k):                                  # same pass: busy SIMD-cycles / elapsed SIMD-cycles at the clock the launch really ran at
        out["per_kernel"][k]["mfma_busy_frac_at_actual_clock"] = round(c / (32.0 * sq[k]), 4)
        out["per_kernel"][k]["shader_clock_ghz_in_pmc_pass"] = round(sq[k] / n / 32.0 / dur[k], 3)
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["per_kernel"].items():
    print(f"{v['mfma_busy_frac']:.3f}  {v['avg_duration_us']:8.1f}us  {k[:80]}")
