#!/bin/bash
# round 6, experiment 4c: fused power iterations with 2 (shipped build) and 1 (lab build, -DMTD_SN_FUSE_GR=1) rows per thread at a time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_discriminator_gpu.py tests/test_generator_gpu.py -x -q -k "power or deferred" > $O/exp5_tests.log 2>&1 || { tail -40 $O/exp5_tests.log; exit 1; }
tail -2 $O/exp5_tests.log
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_SN_FUSED_ITERS=0" "MTD_LAB_LIB=0 MTD_SN_FUSED_ITERS=1" 2 | tee $O/exp5_ab.txt
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_SN_FUSED_ITERS=0" "MTD_SN_FUSED_ITERS=1" 2 | tee -a $O/exp5_ab.txt
