#!/bin/bash
# A/B of the Fresnel column pass (PPV_DFFT_COLS 0 / 1 / 2) under rocprofv3 kernel stats: average duration of the column kernel and the
# camera-alone throughput (tools/bench_camera.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 0 1 2; do
  mkdir -p gpurun_out/dfft_$m
  PPV_DFFT_COLS=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dfft_$m -o s -- python tools/bench_camera.py > gpurun_out/dfft_$m/bench.log 2>&1
  f=$(find gpurun_out/dfft_$m -name "*kernel_stats.csv" | head -1)
  python - "$f" $m <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dfft_cols" in r["Name"]:
        print(f"mode {sys.argv[2]}: {r['Name'][:40]} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us")
PY
  grep -h "^{" gpurun_out/dfft_$m/bench.log | tail -1 | cut -c1-220
  rm -rf gpurun_out/dfft_$m
done
