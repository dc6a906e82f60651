#!/bin/bash
# round 6, experiment 5: radix-8 LDS passes of the whole-slice transforms; fused power iterations with 1024-thread workgroups (lab build);
# the price of the Winograd conv's split of K
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_inference_gpu.py -x -q > $O/exp6_tests.log 2>&1 || { tail -40 $O/exp6_tests.log; exit 1; }
tail -2 $O/exp6_tests.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --workload inference512 --steps 20 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print('inference512 radix-8', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done | tee $O/exp6_inf.txt
MTD_LAB=1 timeout -k 10 300 python tools/splitk_price.py | tee $O/splitk_price.txt
bash tools/ab_step.sh "MTD_LAB_LIB=0" "MTD_X=1" 2 | tee $O/exp6_ab.txt
