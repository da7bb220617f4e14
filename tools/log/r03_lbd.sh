cd $GRAFT_REPO_ROOT
python tools/dbg_lbd.py 2>&1 | grep -v 'rel x 0.00e+00' | tail -12
timeout 600 python -m pytest tests/test_gpu_lbfgs.py -q -x 2>&1 | tail -6
for d in 0 1; do
SPECINV_LBFGS_DEVICE=$d python bench.py --workload C5 --outer 10 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 device=$d', round(d['ms_per_step'],3), 'objective launch_ms', round(d['roofline']['launch_ms'],4), d['check']['ok'], d['config']['lbfgs'])"
done
