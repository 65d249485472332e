#!/bin/bash
# Kernel trace of the DEFAULT (two-stream, overlapped) bench step: per-kernel stats + the timeline of the last step (tools/step_timeline.py).
# usage (GPU box, repo root): bash tools/trace_step.sh <tag>   -> gpurun_out/<tag>_overlapped_kernel_stats.csv, gpurun_out/<tag>_timeline.json
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-trace}; O=$R/gpurun_out/$T.d; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dense --no-roofline > $O/bench.log 2>&1
cd $R
cp $(find $O -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_overlapped_kernel_stats.csv
python3 tools/step_timeline.py $(find $O -name "*kernel_trace.csv" | head -1) gpurun_out/${T}_timeline.json 1
rm -rf $O
