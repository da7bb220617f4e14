cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_lbfgs.py tests/test_gpu_bench_sizes.py tests/test_gpu_two_process.py -q -m gpu 2>&1 | tail -25 > gpurun_out/r04_e_tests.txt
cat gpurun_out/r04_e_tests.txt
for v in baseline wolfe memory; do python3 bench.py --workload C5 --c5-variant $v --steps 3 --warmup 1 --no-cpu-baseline $( [ $v = memory ] && echo --outer 8 ) > gpurun_out/r04_bench_C5_$v.json 2> gpurun_out/r04_bench_C5_$v.err; python3 -c "
import json
d=json.loads(open('gpurun_out/r04_bench_C5_$v.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$v', d['value'], d['ms_per_step'], r['launch_ms'], r.get('frac'), r.get('lds_conflict_frac'), r['hbm'].get('traffic_over_algorithmic'), d['config']['lbfgs'])
"; done
