#!/bin/bash
# round 6, experiment 4b: fused power iterations (alignment fix), forced-DP one-rank line with the N > 1 diagnostics
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
NOX="--no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api --no-wino-split"
timeout -k 10 400 python -m pytest tests/test_discriminator_gpu.py -x -q -k "power or spectral" > $O/exp4_tests.log 2>&1 || { tail -40 $O/exp4_tests.log; exit 1; }
tail -2 $O/exp4_tests.log
bash tools/ab_step.sh "MTD_SN_FUSED_ITERS=0" "MTD_SN_FUSED_ITERS=1" 2 | tee $O/exp4_ab.txt
MTD_FORCE_DP=1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 $NOX > $O/fdp.json 2> $O/fdp.err || { tail -20 $O/fdp.err; exit 1; }
python - <<'PY'
import json
z = json.loads(open("gpurun_out/fdp.json").read().strip().splitlines()[-1])
print({k: z.get(k) for k in ("ms_per_step", "ranks_seen", "allreduce_payload_mb", "allreduce_standalone_ms", "allreduce_alg_gbs", "allreduce_bus_gbs", "launch_mode_per_rank", "ms_per_step_collectives_stubbed", "comm_exposed_ms", "graph_error")})
PY
timeout -k 10 300 python bench.py --steps 30 --warmup 8 $NOX > $O/plain.json 2> $O/plain.err && python -c "
import json; z=json.loads(open('gpurun_out/plain.json').read().strip().splitlines()[-1]); print('plain', z['ms_per_step'])"
