#!/bin/bash
# Same-box A/B of the non-temporal output stores: conv kernels (PPV_NT_STORE bits: 1 tile epilogue, 2 conv1x1_stream, 4 zero fill of
# conv_dgrad_s2) x element-wise kernels (in-tree lib vs lib_ab built with the other -DPPV_NT_ELT on trunk_ops.hip).
# usage (GPU box, repo root): bash tools/ab_nt.sh "0:ab 1:ab 3:ab 7:ab 1:tree"
AB=/root/repo/privacy-preserving-vision_amd/lib_ab/libppv_hip.so
for i in 1 2; do for c in ${1:-0:tree 1:tree}; do
  nt=${c%%:*}; lib=${c##*:}
  if [ "$lib" = "ab" ]; then export PPV_LIB_PATH=$AB; else unset PPV_LIB_PATH; fi
  PPV_NT_STORE=$nt timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > gpurun_out/bench_nt.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_nt.log") if x.startswith("{")][-1]
d=json.loads(l); print("PPV_NT_STORE=$nt lib=$lib", d["value"], d.get("value_lazy_consumer"), d["ms_per_step"], d.get("windows_ms_per_step"))
PY
done; done
