cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_lbfgs.py tests/test_gpu_bench_sizes.py -q -x -k "not c2_headline and not c4 and not c3" 2>&1 | tail -3
for i in 1 2; do
python bench.py --workload C5 --outer 10 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5', round(d['ms_per_step'],3), 'objective launch_ms', round(d['roofline']['launch_ms'],4), d['roofline']['launches_timed'], d['check']['ok'], round(d['value']/1e6,2))"
done
