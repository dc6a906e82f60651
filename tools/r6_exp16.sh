#!/bin/bash
# round 6, final checks: the GPU suite + smoke at the final tree, the forced data-parallel line beside the plain one, the two-rank rehearsal
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.log 2>&1; echo "gpu suite exit $?"; tail -3 $O/gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
for cfg in "MTD_FORCE_DP=0" "MTD_FORCE_DP=1" "MTD_FORCE_DP=0" "MTD_FORCE_DP=1"; do
  env $cfg timeout -k 10 300 python bench.py --steps 30 --warmup 8 $NOX 2>/dev/null > $O/fdp_$cfg.json
  python -c "
import sys,json; z=json.loads(open('$O/fdp_$cfg.json').read().strip().splitlines()[-1]); print('[$cfg]', z['ms_per_step'], z.get('ms_per_step_collectives_stubbed'), z.get('comm_exposed_ms'), z.get('graph_error'))"
done | tee $O/exp16_fdp.txt
bash tools/dp_two_ranks.sh 2>&1 | grep "dp2\|exit code" | cut -c1-400
