python -m pytest tests/test_kernels_gpu.py -k "block_tail" -x -q 2>&1 | tail -15 && python -m pytest tests/test_generator_gpu.py -x -q 2>&1 | tail -5 && python bench.py --workload generator --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('GEN ms', d['ms_per_step'], 'frac', d.get('step_frac_of_fp32_mfma_peak'))
r=d['roofline']; print(r['kernel'], r['avg_launch_us'], r.get('other_mfma_kernels'))
"
