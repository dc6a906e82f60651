#!/bin/bash
# round 6: the GPU suite at the current defaults; four A/B pairs default vs single-stream Winograd block forms; the two-rank rehearsal
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests_b.log 2>&1; echo "gpu suite exit $?"; tail -3 $O/gpu_tests_b.log
bash tools/ab_step.sh "MTD_LAB_LIB=0" "MTD_LAB_LIB=0 MTD_BLOCK_FWD_WINO=2 MTD_BLOCK_BWD_WINO=3" 4 | tee $O/exp9_ab.txt
bash tools/ab_step.sh "MTD_LAB_LIB=0" "MTD_LAB_LIB=0 MTD_BLOCK_BWD_WINO=3" 2 | tee -a $O/exp9_ab.txt
bash tools/dp_two_ranks.sh 2>&1 | tail -12 | cut -c1-700
