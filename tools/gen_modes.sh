#!/bin/bash
# generator workload under its three launch modes + the replay tests
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_generator_gpu.py -m gpu -x -q -k "workload or deferred" > $O/pytest_modes.log 2>&1 || { tail -40 $O/pytest_modes.log; exit 1; }
tail -3 $O/pytest_modes.log
for m in list 1 0; do
  MTD_GRAPH=$m timeout -k 10 200 python bench.py --workload generator --no-cpu-baseline --no-roofline > $O/bench_gen_$m.json 2> $O/bench_gen_$m.err
  python - <<P
import json; d=json.loads(open("$O/bench_gen_$m.json").read().strip().splitlines()[-1]); print("$m", d["ms_per_step"], d["value"], d["launch_mode"], d["graph_error"])
P
done
