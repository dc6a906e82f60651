#!/bin/bash
# round 6, experiment 2: activation-side <G, W> of the spectral-norm correction; Winograd block variants in the full step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_discriminator_gpu.py tests/test_step_gpu.py -x -q > $O/exp2_tests.log 2>&1 || { tail -40 $O/exp2_tests.log; exit 1; }
tail -2 $O/exp2_tests.log
bash tools/ab_step.sh "MTD_SN_ACT_DOT=0" "MTD_SN_ACT_DOT=1" 2 | tee $O/exp2_ab.txt
bash tools/ab_step.sh "MTD_BLOCK_BWD_WINO=1" "MTD_BLOCK_BWD_WINO=1 MTD_BLOCK_FWD_WINO=1" 2 | tee -a $O/exp2_ab.txt
