#!/bin/bash
# round 6, experiment 9: the two decoders' mirror convs in pairs (kernels.conv_pair / wino_conv_multi_kernel)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_pair or winograd_conv_vs_torch" > $O/exp13_tests.log 2>&1 || { tail -40 $O/exp13_tests.log; exit 1; }
tail -2 $O/exp13_tests.log
timeout -k 10 600 python -m pytest tests/test_discriminator_gpu.py tests/test_step_gpu.py -x -q > $O/exp13_tests_step.log 2>&1 || { tail -40 $O/exp13_tests_step.log; exit 1; }
tail -2 $O/exp13_tests_step.log
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_PAIR_DECODERS=0" "MTD_LAB_LIB=0 MTD_PAIR_DECODERS=1" 3 | tee $O/exp13_ab.txt
