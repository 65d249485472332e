#!/bin/bash
# Memory-path counters per kernel of one bench step (separate rocprofv3 --pmc passes, never combined with tracing): where does a
# fetch-bound MFMA kernel wait?  (No TA_* pass: rocprofv3 aborted with signal 6 on "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES ..." here.)  usage (GPU box, repo root): bash tools/pmc_study.sh   -> gpurun_out/pmc_study/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_study
rm -rf $O; mkdir -p $O
export PPV_WGRAD_SIDE=0
i=0
for set in "TCC_HIT TCC_MISS TCC_REQ TCC_READ" \
           "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" \
           "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_STALL_MULTI_MISS" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES TCP_LFIFO_STALL_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o c -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/p$i.log 2>&1
  echo "pass $i done: $set"
done
python tools/pmc_study.py $O > $O/summary.txt
rm -rf $O/p[0-9]
cat $O/summary.txt
