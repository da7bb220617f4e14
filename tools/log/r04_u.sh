cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for g in "" "--generic"; do
python3 bench.py --workload C1 $g --no-extra --no-pmc --no-h2d --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C1 $g', round(d['value']/1e6,2), round(d['ms_per_step'],4), d['check'].get('ok') if d.get('check') else None, d.get('roofline',{}).get('launch_ms'), d['config'].get('kernel_path'))"
done
