cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|Error" | head -10
for v in 1 0; do
SPECINV_LBFGS_DEVICE_WOLFE=$v python3 bench.py --workload C5 --c5-variant wolfe --no-extra --no-pmc 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 wolfe device=$v', round(d['value']/1e6,2), round(d['ms_per_step'],3), d['check']['ok'], d.get('roofline',{}).get('launch_ms'), d['config'].get('lbfgs'))"
done
SPECINV_LBFGS_DEVICE_WOLFE=1 python3 bench.py --workload C5 --c5-variant memory --no-extra --no-pmc 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 memory', round(d['value']/1e6,2), round(d['ms_per_step'],3), d['check']['ok'], d.get('roofline',{}).get('launch_ms'), d['config'].get('lbfgs'))"
