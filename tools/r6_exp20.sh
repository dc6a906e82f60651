#!/bin/bash
# round 6, experiment 14 (lab library): three passes advanced together with the groups planned as pairs (MTD_WINO_PAIR_SPLIT=2) against the shipped two
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
for rep in 1 2 3; do
for cfg in "MTD_LAB_LIB=0" "MTD_LOCKSTEP_PASSES=3 MTD_WINO_PAIR_SPLIT=2" "MTD_LOCKSTEP_PASSES=3"; do
  ms=$(env MTD_LAB=1 $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [$cfg] $ms ms"
done
done | tee $O/exp20_ab.txt
