#!/bin/bash
# round 6, experiment 11: pair launches only up to a map size (MTD_PAIR_MAX_PIXELS)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
for rep in 1 2; do
for px in 0 4096 16384 65536; do
  ms=$(env MTD_LAB=1 MTD_LAB_LIB=0 MTD_PAIR_MAX_PIXELS=$px timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "step [MTD_PAIR_MAX_PIXELS=$px] $ms ms"
done
done | tee $O/exp15_ab.txt
