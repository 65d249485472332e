for i in 1 2 3; do for h in 0 1; do PPV_HALO64=$h timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_h64_$h.log 2>&1; python - <<PY
import json
l=[x for x in open("gpurun_out/bench_h64_$h.log") if x.startswith("{")][-1]
d=json.loads(l); print("halo64=$h", d["value"], d.get("value_dense_surface"), d["ms_per_step"])
PY
done; done
