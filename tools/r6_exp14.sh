#!/bin/bash
# round 6, experiment 10: the adversarial and the first consistency pass advanced together (discriminator_path.disc_backward_lockstep)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_discriminator_gpu.py -x -q > $O/exp14_tests.log 2>&1 || { tail -40 $O/exp14_tests.log; exit 1; }
tail -2 $O/exp14_tests.log
bash tools/ab_step.sh "MTD_LAB_LIB=0 MTD_LOCKSTEP_PASSES=0" "MTD_LAB_LIB=0 MTD_LOCKSTEP_PASSES=1" 3 | tee $O/exp14_ab.txt
