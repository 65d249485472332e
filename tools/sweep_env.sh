#!/bin/bash
# A/B of environment switches through the real bench step on ONE box: each argument is a quoted "VAR=val VAR2=val" set ("-" = none).
# Two passes over the list so that a drift of the box shows.  usage: tools/sweep_env.sh out.log "-" "PPV_X=1" ...
out=$1; shift
: > "$out"
for pass in 1 2; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    line=$(env $envs python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-dense --no-roofline 2>/dev/null | tail -1)
    v=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['value'], d['ms_per_step'])" "$line" 2>/dev/null)
    echo "pass $pass  [$cfg]  $v" | tee -a "$out"
  done
done
