#!/bin/bash
# In-step duration of every launch shape (every launch alone on the device): kernel trace of a short bench run -> tools/step_by_shape.py
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/shapes; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export PPV_WGRAD_SIDE=0
rocprofv3 --kernel-trace --output-format csv -d $O/t -o s -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dense --no-roofline --no-configs > $O/run.log 2>&1
python3 $R/tools/step_by_shape.py $(find $O/t -name "*kernel_trace.csv" | head -1) ${1:-100} > $O/shapes.txt
rm -rf $O/t
