#!/bin/bash
# round 6, experiment 12b: the pair launches' split of K planned for 2 (shipped), 3, 4 problems' worth of grid (lab library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
for rep in 1 2; do
for v in 2 3 4 1; do
  ms=$(env MTD_LAB=1 MTD_WINO_PAIR_SPLIT=$v timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [MTD_WINO_PAIR_SPLIT=$v] $ms ms"
done
done | tee $O/exp17b_ab.txt
