#!/bin/bash
# round 6, experiments 3 + 4: the plain 32 -> 32 layers' data gradients on the persistent F(2x4) kernel's MASKED2 form; fused power iterations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_discriminator_gpu.py -x -q -k "power or spectral or sn_grad or train_forward" > $O/exp4_tests.log 2>&1 || { tail -40 $O/exp4_tests.log; exit 1; }
tail -2 $O/exp4_tests.log
timeout -k 10 400 python -m pytest tests/test_step_gpu.py -x -q > $O/exp4_tests_step.log 2>&1 || { tail -40 $O/exp4_tests_step.log; exit 1; }
tail -2 $O/exp4_tests_step.log
MTD_LAB=1 MTD_WINO_C32_BWD=1 timeout -k 10 300 python -m pytest tests/test_generator_gpu.py -x -q -k "oracle or full_batch or replay" > $O/exp3_tests_gen.log 2>&1 || { tail -40 $O/exp3_tests_gen.log; exit 1; }
tail -2 $O/exp3_tests_gen.log
bash tools/ab_step.sh "MTD_SN_FUSED_ITERS=0" "MTD_SN_FUSED_ITERS=1" 2 | tee $O/exp4_ab.txt
bash tools/ab_gen.sh "MTD_X=0" "MTD_WINO_C32_BWD=1" 2 | tee $O/exp3_ab.txt
bash tools/ab_gen.sh "MTD_WINO_C32_BWD=1 MTD_BLOCK_BWD_WINO=1 MTD_BLOCK_FWD_WINO=1" "MTD_X=0" 1 | tee -a $O/exp3_ab.txt
bash tools/ab_step.sh "MTD_X=0" "MTD_WINO_C32_BWD=1" 2 | tee -a $O/exp3_ab.txt
