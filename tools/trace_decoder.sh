#!/bin/bash
# kernel stats of the config-3 step (bench.py --decoder): decoder families per step under PPV_DEC_WGRAD=lib / x3 (default) / hip
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/dec.d; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for mode in ${MODES:-lib x3}; do
  export PPV_DEC_WGRAD=$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o s -- python3 $R/bench.py --decoder --steps 4 --warmup 2 --no-cpu-baseline --no-dense --no-roofline > $O/$mode.log 2>&1
  echo "== PPV_DEC_WGRAD=$mode: $(tail -1 $O/$mode.log | cut -c1-120)"
  python3 $R/tools/decoder_kernels.py $(find $O/$mode -name "*kernel_stats.csv" | head -1) 6
  cp $(find $O/$mode -name "*kernel_stats.csv" | head -1) $R/gpurun_out/dec_kernel_stats_$mode.csv
done
rm -rf $O
