#!/bin/bash
# round 6, experiment 17: the restoration pass's trunk advanced together with the second consistency pass's (MTD_LOCKSTEP_PASSES=2)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_step_gpu.py -x -q -k "paired_launches" > $O/exp23_tests.log 2>&1 || { tail -30 $O/exp23_tests.log | cut -c1-300; exit 1; }
tail -2 $O/exp23_tests.log
bash tools/ab_step.sh "MTD_LOCKSTEP_PASSES=1" "MTD_LOCKSTEP_PASSES=2" 3 | tee $O/exp23_ab.txt
