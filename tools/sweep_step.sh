#!/bin/bash
# In-step sweep of lab switches on one box: tools/sweep_step.sh "CFG1" "CFG2" ...  (each: one ENV=VALUE string; the first should be a no-op
# such as MTD_X=1; every configuration is run `PAIRS` times, interleaved)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split --no-live-pmc"
for i in $(seq ${PAIRS:-2}); do
  for cfg in "$@"; do
    ms=$(env MTD_LAB=1 $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "[$cfg] $ms"
  done
done
