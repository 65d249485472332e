#!/bin/bash
# Same-box A/B of one environment variable over a LIST of values on the headline step (lean: no roofline / counters / side configs):
# N alternating rounds of `bench.py --steps 20 --warmup 5`; prints value (dense surface), the lazy-consumer value and the windows.
# usage (GPU box, repo root): bash tools/ab_vals.sh PPV_WGRAD_FORKS "0 1 2" [rounds]
V=$1; VALS=$2; N=${3:-2}
for i in $(seq 1 $N); do for h in $VALS; do
  env $V=$h timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-configs --no-live-pmc > gpurun_out/bench_ab_$h.log 2>gpurun_out/bench_ab_$h.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_ab_$h.log") if x.startswith("{")]
if not l:
    print("$V=$h: no JSON line"); print(open("gpurun_out/bench_ab_$h.err").read()[-800:])
else:
    d=json.loads(l[-1]); print("$V=$h dense", d["value"], d["windows_ms_per_step"], "lazy", d.get("value_lazy_consumer"), d.get("windows_ms_per_step_other_surface"))
PY
done; done
