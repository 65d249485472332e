#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats of a B=128 bench run + the two PMC passes, summaries under
# gpurun_out/prof_round/ (copy what is to be judged into profiles/).  The program itself follows `--` (no env / bash -c hops).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_round
mkdir -p $O
export PPV_WGRAD_SIDE=0   # per-kernel durations and counters are taken with every launch alone on the device
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense > $O/bench_under_rocprof.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-roofline > $O/pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-roofline > $O/pmc_write.log 2>&1
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/mfma -o m -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dense --no-roofline > $O/pmc_mfma.log 2>&1
echo mfma done
python tools/collect_pmc.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python tools/collect_mfma.py $(find $O/mfma -name "*counter_collection.csv" | head -1) $O/kernel_stats.csv $O/mfma_util.json | head -12
rm -rf $O/fetch $O/write $O/stats $O/mfma
unset PPV_WGRAD_SIDE
# optional tag (e.g. r03f): the fresh summaries go into profiles/ of THIS copy first, so that the bench line below reads them
# (its counter fields are "stale" whenever the sources' hash differs from the one stored in the profile)
if [ -n "$1" ]; then
  cp $O/kernel_stats.csv profiles/$1_bench_b128_kernel_stats.csv
  cp $O/pmc_traffic.json profiles/$1_pmc_traffic.json
  cp $O/mfma_util.json profiles/$1_mfma_util.json
fi
python bench.py > $O/bench_default.json.log 2>$O/bench_default.err
tail -1 $O/bench_default.json.log | cut -c1-300
