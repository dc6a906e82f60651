#!/bin/bash
# halo-tile conv lab switches: two-workgroup variant (MTD_C32T_VARIANT=1, MTD_C32T_STAGGER) and non-temporal output stores
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
MTD_IGEMM_NT=1 timeout -k 10 300 python -m pytest tests/test_generator_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "generator or conv_fwd" > $O/pytest_c32u.log 2>&1 || { tail -30 $O/pytest_c32u.log; exit 1; }
tail -1 $O/pytest_c32u.log
for cfg in "0 0 0" "0 0 1" "1 0 1" "1 32 1"; do
  set -- $cfg
  MTD_C32T_VARIANT=$1 MTD_C32T_STAGGER=$2 MTD_IGEMM_NT=$3 timeout -k 10 200 python bench.py --workload generator --no-cpu-baseline > $O/bench_c32_$1_$2_$3.json 2> $O/bench_c32_$1_$2_$3.err
  python - <<P
import json; d=json.loads(open("$O/bench_c32_$1_$2_$3.json").read().strip().splitlines()[-1]); print("variant $1 stagger $2 nt $3:", d["ms_per_step"], "ms", d["roofline"]["avg_launch_us"], "us/launch")
P
done
MTD_IGEMM_NT=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline > $O/bench_full_nt.json 2> $O/bench_full_nt.err
python - <<P
import json; d=json.loads(open("$O/bench_full_nt.json").read().strip().splitlines()[-1]); print("full step nt 1:", d["ms_per_step"], "ms")
P
