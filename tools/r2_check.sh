#!/bin/bash
# Round-2 GPU-box pass: parity tests (all failures shown), the bench line in both profiler timing modes, and the
# single-stream rocprofv3 kernel tables the roofline figures are compared with.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 700 python -m pytest tests -m gpu -q --maxfail=12 > $O/pytest_gpu.log 2>&1; fi
echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_full.json 2> $O/bench_full.err || { echo "bench failed"; tail -20 $O/bench_full.err; exit 1; }
cut -c1-1500 $O/bench_full.json
MTD_PROF_MODE=bracket timeout -k 10 200 python bench.py --no-cpu-baseline --no-generator > $O/bench_full_bracket.json 2> $O/bench_full_bracket.err || { echo "bench bracket failed"; exit 1; }
python - <<'PY'
import json
for f in ("bench_full", "bench_full_bracket"):
    z = json.load(open(f"gpurun_out/{f}.json"))
    r = z["roofline"]
    print(f, z["ms_per_step"], r["kernel"], r["avg_launch_us"], r["frac"], r["timing"][:40])
PY
rm -rf $O/prof_full $O/prof_gen
MTD_NO_SIDE_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_full -o full -- python3 bench.py --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-generator > $O/prof_full.log 2>&1 || { echo "rocprof full failed"; tail -5 $O/prof_full.log; exit 1; }
MTD_NO_SIDE_STREAMS=1 MTD_GRAPH=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_gen -o gen -- python3 bench.py --workload generator --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/prof_gen.log 2>&1 || { echo "rocprof gen failed"; exit 1; }
find $O/prof_gen $O/prof_full -name "*.db" | while read f; do python tools/rocpd_stats.py $f ${f%.db}_kernel_stats.csv --steps $( [[ $f == *gen* ]] && echo 13 || echo 7 ); done
find $O/prof_gen $O/prof_full -name "*.db" -size +30M -delete
find $O/prof_full $O/prof_gen -name "*kernel_stats.csv" | head
