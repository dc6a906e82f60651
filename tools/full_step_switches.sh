#!/bin/bash
# A/B of the generator-pass switches inside the full step (eager, multi-stream)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for sw in "none" "MTD_NO_DEFERRED_WGRAD" "MTD_NO_FUSED_ACT_GRAD"; do
  env $( [ $sw = none ] && echo "X_=1" || echo "$sw=1" ) timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full step, switch $sw rep $rep:', d['ms_per_step'], 'ms')"
done; done
