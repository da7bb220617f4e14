cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_fast.py tests/test_gpu_semi.py tests/test_gpu_properties.py tests/test_gpu_bench_sizes.py tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_random_configs.py -q -m gpu -x 2>&1 | tail -15)
for w in C4 C2; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > gpurun_out/t_bench_$w.json 2> gpurun_out/t_bench_$w.err
  python -c "
import json
d=json.loads(open('gpurun_out/t_bench_$w.json').read().strip().splitlines()[-1])
print('$w', round(d['ms_per_step'],3), '%.4g'%d['value'], 'launch_ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'check', d['check'])"
done
