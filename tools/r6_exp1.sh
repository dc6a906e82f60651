#!/bin/bash
# round 6, experiment 1: Res-FFT block conv on the persistent F(2x4) kernel (forward / data gradient) -- parity under the switches, then A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
MTD_LAB=1 MTD_BLOCK_BWD_WINO=1 MTD_BLOCK_FWD_WINO=1 timeout -k 10 300 python -m pytest tests/test_generator_gpu.py -x -q > $O/exp1_tests.log 2>&1 || { tail -30 $O/exp1_tests.log; exit 1; }
tail -2 $O/exp1_tests.log
bash tools/ab_gen.sh "MTD_X=0" "MTD_BLOCK_BWD_WINO=1" 2 | tee $O/exp1_ab.txt
bash tools/ab_gen.sh "MTD_BLOCK_FWD_WINO=1" "MTD_BLOCK_BWD_WINO=1 MTD_BLOCK_FWD_WINO=1" 2 | tee -a $O/exp1_ab.txt
