#!/bin/bash
# PMC passes (one counter group per run, kernel trace only) over tools/c32_conv_time.py: fabric bytes, L2 hit rate and where the
# waves of the persistent 32 -> 32 channel Winograd kernel spend their cycles.  Usage (on the box): bash tools/c32_pmc.sh [B H W]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "valu:SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "wait:SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rm -rf $O/c32pmc_$name
  timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/c32pmc_$name -- python3 tools/c32_conv_time.py "$@" > $O/c32pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/c32pmc_$name.log; continue; }
  python tools/pmc_summary.py $O/c32pmc_$name $O/c32pmc_$name.csv
  python - $O/c32pmc_$name.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "wino_c32" in r["Kernel_Name"]:
        print(r["Kernel_Name"].split("::")[-1][:40], r["Counter_Name"], r["AvgPerDispatch"], r["Dispatches"])
PY
  rm -rf $O/c32pmc_$name
done
