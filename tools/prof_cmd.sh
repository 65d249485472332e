#!/bin/bash
# kernel-stats of an arbitrary python tool: tools/prof_cmd.sh <tag> <script.py> [args]; prints the top kernels by total time
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; shift; O=$R/gpurun_out/$T.d; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 $R/$@ > $O/run.log 2>&1
cd $R
python3 - <<PY
import csv,glob,re
f=glob.glob('$O/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:${TOPN:-25}]:
    name = re.sub(r"[(].*", "", r["Name"]).replace("void ", "")[:100]
    print("%10.1f us %6d x %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e3, int(r["Calls"]), float(r["AverageNs"]) / 1e3, name))
PY
rm -rf $O
