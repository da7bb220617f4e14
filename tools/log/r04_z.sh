cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python3 -m pytest tests/test_gpu_lbfgs.py -q -x -k "wolfe" 2>&1 | tail -3
for v in 1 0 1; do
SPECINV_LBFGS_DEVICE_WOLFE=$v timeout 120 python3 bench.py --workload C5 --c5-variant wolfe --no-extra --no-pmc --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 wolfe device=$v', round(d['value']/1e6,2), round(d['ms_per_step'],3), d['check']['ok'], d.get('roofline',{}).get('launch_ms'))"
done
for v in 1 0; do
SPECINV_LBFGS_DEVICE_WOLFE=$v timeout 200 python3 bench.py --workload C5 --c5-variant memory --steps 2 --warmup 1 --no-extra --no-pmc --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 memory device=$v', round(d['value']/1e6,2), round(d['ms_per_step'],3), d['check']['ok'], d['config']['lbfgs']['evaluations'], d['config']['lbfgs']['inner_iterations'])"
done
