#!/bin/bash
# PMC passes on the vector-memory path (address unit TA, L1 TCP, data return TD) of the persistent 32 -> 32 channel Winograd kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for pass in "ta1:TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "ta2:TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "ta3:TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" "tcp1:TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "tcp2:TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "tcp3:TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_LATENCY_sum" "td:TD_TD_BUSY_sum TD_TC_STALL_sum" "sqv:SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf $O/c32pmc_$name
  timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/c32pmc_$name -- python3 tools/c32_conv_time.py "$@" > $O/c32pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $O/c32pmc_$name.log; continue; }
  python tools/pmc_summary.py $O/c32pmc_$name $O/c32pmc_$name.csv
  python - $O/c32pmc_$name.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "wino_c32" in r["Kernel_Name"]:
        print(r["Kernel_Name"].split("::")[-1][:40], r["Counter_Name"], r["AvgPerDispatch"], r["Dispatches"])
PY
  rm -rf $O/c32pmc_$name
done
