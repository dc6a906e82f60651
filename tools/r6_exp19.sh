#!/bin/bash
# round 6: the GPU suite + smoke at the final tree, then the measurement pipeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.log 2>&1; rc=$?; echo "gpu suite exit $rc"; tail -3 $O/gpu_tests_final.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/measure.sh r6 > $O/measure_final.log 2>&1; tail -3 $O/measure_final.log | cut -c1-300
