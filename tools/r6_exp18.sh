#!/bin/bash
# round 6, experiment 13b: passes advanced together -- none (0), two (2), three (1) -- on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
for rep in 1 2 3; do
for v in 0 2 1; do
  ms=$(env MTD_LAB=1 MTD_LAB_LIB=0 MTD_LOCKSTEP_PASSES=$v timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [MTD_LOCKSTEP_PASSES=$v] $ms ms"
done
done | tee $O/exp18b_ab.txt
