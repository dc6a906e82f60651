#!/bin/bash
# A/B of the generator leg (BASELINE configs[1]) on one box: tools/ab_gen.sh "ENV_A" "ENV_B" [pairs]   (MTD_LAB=1 for both)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A="$1"; B="$2"; N="${3:-2}"
for i in $(seq $N); do
  for cfg in "$A" "$B"; do
    ms=$(env MTD_LAB=1 $cfg timeout -k 10 200 python bench.py --workload generator --steps 40 --warmup 10 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "[$cfg] $ms ms"
  done
done
