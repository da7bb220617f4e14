cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_lbfgs.py -q -k "one_launch" 2>&1 | grep -E "passed|failed|^FAILED|Error" | head -5
for i in 1 2; do
python3 bench.py --workload C5 --no-extra --no-pmc 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5', round(d['value']/1e6,2), round(d['ms_per_step'],2), d['check']['ok'], d.get('roofline',{}).get('launch_ms'))"
done
