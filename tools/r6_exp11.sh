#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 200 python tools/r6_probe_routes.py 2>&1 | grep -v "amdgpu.ids" | tee $O/exp11_routes.txt
MTD_LAB=1 MTD_LAB_LIB=0 MTD_BLOCK_FWD_WINO=0 MTD_BLOCK_BWD_WINO=0 timeout -k 10 200 python tools/r6_probe_routes.py 2>&1 | grep -v "amdgpu.ids" | tee -a $O/exp11_routes.txt
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_SN_MERGE_HALVES=0" "MTD_LAB_LIB=0 MTD_SN_MERGE_HALVES=1" 3 | tee $O/exp10_ab.txt
