cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_lbfgs.py tests/test_gpu_api.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | head -20
for v in 1 0; do
SPECINV_OBJ_SPARSE=$v python3 bench.py --workload C5 --no-extra --no-pmc 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 sparse=$v', round(d['value']/1e6,2), round(d['ms_per_step'],2), d['check']['ok'], d.get('roofline',{}).get('launch_ms'))"
done
