#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_generator_gpu.py tests/test_kernels_gpu.py -m gpu -x -q > $O/pytest_quick.log 2>&1 || { tail -40 $O/pytest_quick.log; exit 1; }
tail -2 $O/pytest_quick.log
timeout -k 10 200 python bench.py --workload generator --no-cpu-baseline > $O/bench_gen_quick.json 2> $O/bench_gen_quick.err
python - <<P
import json; d=json.loads(open("$O/bench_gen_quick.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"], d["launch_mode"], d["graph_error"], d["roofline"]["avg_launch_us"], d["step_frac_of_fp32_mfma_peak"])
P
