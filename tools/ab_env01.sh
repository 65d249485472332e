#!/bin/bash
# Same-box A/B of one 0/1 environment switch on the headline step: three alternating pairs of `bench.py --steps 20 --warmup 5`.
# usage (GPU box, repo root): bash tools/ab_env01.sh PPV_HALO64
V=${1:-PPV_HALO64}
for i in 1 2 3; do for h in 0 1; do
  env $V=$h timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_ab_$h.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_ab_$h.log") if x.startswith("{")][-1]
d=json.loads(l); print("$V=$h", d["value"], d.get("value_lazy_consumer"), d["ms_per_step"], d.get("windows_ms_per_step"))
PY
done; done
