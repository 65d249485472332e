#!/bin/bash
# Kernel trace of the default (two-stream) bench step; keeps the trace csv under gpurun_out/<tag>_trace.csv for tools/step_window.py
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-tw}; O=$R/gpurun_out/$T.d; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o s -- python3 $R/bench.py --steps 3 --warmup 2 --lazy --no-cpu-baseline --no-dense --no-roofline --no-configs --no-live-pmc > $O/bench.log 2>&1
cd $R
F=$(find $O -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $F gpurun_out/${T}_timeline.json 1
python3 tools/step_window.py $F ${2:-9000} ${3:-10200} > gpurun_out/${T}_window.txt
python3 tools/step_window.py $F 0 100000 > gpurun_out/${T}_all.txt
rm -rf $O
