#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
for pad in 0 65536 40000; do
  echo "== MTD_WGRAD_LDS_PAD=$pad"
  MTD_WGRAD_LDS_PAD=$pad WL=generator TOP=3 MTD_GRAPH=0 timeout -k 10 200 python tools/shape_table.py 2>&1 | grep -v amdgpu.ids
  MTD_WGRAD_LDS_PAD=$pad timeout -k 10 200 python bench.py --workload generator --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('generator step', d['ms_per_step'], 'ms')"
done
MTD_WGRAD_LDS_PAD=65536 timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full step pad 65536', d['ms_per_step'], 'ms')"
