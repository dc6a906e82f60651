#!/bin/bash
# A/B of the full training step on one box: tools/ab_step.sh "ENV_A" "ENV_B" [pairs]   (MTD_LAB=1 is set for both; the lab library is loaded when it exists)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
A="$1"; B="$2"; N="${3:-2}"
for i in $(seq $N); do
  for cfg in "$A" "$B"; do
    ms=$(env MTD_LAB=1 $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "[$cfg] $ms ms"
  done
done
