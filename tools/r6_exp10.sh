#!/bin/bash
# round 6, experiment 8: both halves of a paired pass in one small-map weight-gradient launch (mtd_wgrad_args.half_scale)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_discriminator_gpu.py -x -q -k "half_scale or sn_grad or weight_gradient_pair or train_forward or spectral" > $O/exp10_tests.log 2>&1 || { tail -40 $O/exp10_tests.log; exit 1; }
tail -2 $O/exp10_tests.log
timeout -k 10 500 python -m pytest tests/test_step_gpu.py tests/test_generator_gpu.py -x -q > $O/exp10_tests_step.log 2>&1 || { tail -40 $O/exp10_tests_step.log; exit 1; }
tail -2 $O/exp10_tests_step.log
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_SN_MERGE_HALVES=0" "MTD_LAB_LIB=0 MTD_SN_MERGE_HALVES=1" 3 | tee $O/exp10_ab.txt
