"""Print rocprofv3 kernel_stats.csv rows whose name contains any of the given substrings.  usage: kernel_stats_grep.py <csv> <substr>..."""
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("ppv::", "")
    if any(k in n for k in sys.argv[2:]):
        print(f"{int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:8.1f} us = {float(r['TotalDurationNs']) / 1e6:8.3f} ms  {n[:80]}")
