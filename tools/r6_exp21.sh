#!/bin/bash
# round 6, experiment 15: which early-shipping point costs the forced data-parallel path its 0.34 ms at one rank?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
for rep in 1 2; do
for cfg in "MTD_FORCE_DP=0" "MTD_FORCE_DP=1" "MTD_FORCE_DP=1 MTD_DP_SHIP_STAGES=heads" "MTD_FORCE_DP=1 MTD_DP_SHIP_STAGES=trunk_low" "MTD_FORCE_DP=1 MTD_DP_EARLY_SHIP=0"; do
  ms=$(env MTD_LAB=1 MTD_LAB_LIB=0 $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [$cfg] $ms ms"
done
done | tee $O/exp21_ab.txt
