#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do for pad in 0 65536 100000; do
  MTD_WGRAD_LDS_PAD=$pad timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full step pad $pad rep $rep', d['ms_per_step'], 'ms')"
done; done
