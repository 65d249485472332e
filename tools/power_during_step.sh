#!/bin/bash
# Clocks and power while the headline step runs (is the step power- / clock-limited?): samples rocm-smi every 0.5 s beside bench.py --steps 200.
python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-roofline --no-configs --no-live-pmc > gpurun_out/power_bench.log 2>&1 &
P=$!
sleep 25
for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | tr '\n' ' '; echo; sleep 0.5; done
wait $P
tail -1 gpurun_out/power_bench.log | cut -c1-160
echo "--- idle"; sleep 3; rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr '\n' ' '; echo
