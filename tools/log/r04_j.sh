cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_lbfgs.py tests/test_gpu_bench_sizes.py -q -m gpu -x 2>&1 | tail -4 | cut -c1-200
for i in 1 2; do python3 bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5', round(d['value']/1e6,2),'M', round(d['ms_per_step'],2),'ms obj', round(d['roofline']['launch_ms'],4), d['check']['ok'])"; done
python3 tools/bench_objective.py 2>&1 | tail -12
