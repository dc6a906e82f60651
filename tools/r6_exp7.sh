#!/bin/bash
# round 6, experiment 6: radix-8 (shipped build) against radix-4 (lab build, -DMTD_ANY_R8=0) LDS passes on one box; forced-DP with and without early shipping
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
for i in 1 2 3; do
  for cfg in "MTD_X=0" "MTD_LAB=1"; do
    ms=$(env $cfg timeout -k 10 200 python bench.py --workload inference512 --steps 20 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "inference512 [$cfg: MTD_LAB=1 = radix-4 lab build] $ms ms"
  done
done | tee $O/exp7_inf.txt
for cfg in "MTD_FORCE_DP=0" "MTD_FORCE_DP=1" "MTD_FORCE_DP=1 MTD_LAB=1 MTD_LAB_LIB=0 MTD_DP_EARLY_SHIP=0" "MTD_FORCE_DP=0" "MTD_FORCE_DP=1"; do
  env $cfg timeout -k 10 300 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "
import sys,json; z=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg]', z['ms_per_step'], z.get('ms_per_step_collectives_stubbed'), z.get('comm_exposed_ms'))"
done | tee $O/exp7_fdp.txt
