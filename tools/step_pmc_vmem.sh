#!/bin/bash
# PMC passes on the vector-memory path (TA busy, L1 -> L2 read requests, TD busy) per kernel of the full training step (eager,
# single stream, so that every dispatch is attributed).  Usage (on the box): bash tools/step_pmc_vmem.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp MTD_LAB=1 MTD_GRAPH=0 MTD_NO_SIDE_STREAMS=1 MTD_LIST=0      # (+ MTD_WINO_SPLIT=1 from the caller: the split-bf16 Winograd kernel)
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
for pass in "ta:TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum" "tcp:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "sq:SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf $O/steppmc_$name
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/steppmc_$name -- python3 bench.py --steps 2 --warmup 1 $NOX > $O/steppmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $O/steppmc_$name.log; continue; }
  python tools/pmc_summary.py $O/steppmc_$name $O/steppmc_$name.csv
  rm -rf $O/steppmc_$name
  echo "pass $name done"
done
