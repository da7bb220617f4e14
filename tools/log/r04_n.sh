cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_lbfgs.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | head -20
for w in 0 4; do
  SPECINV_STAMP_LIB=$PWD/spectrogram_inversion_amd/variants/libspecinv_stamps$w.so python3 tools/obj_stamps.py 2>&1 | grep -v "^$" | tail -14
done
python3 bench.py --workload C5 --no-extra --no-pmc 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5', round(d['value']/1e6,2), d['ms_per_step'], d.get('objective_ms_per_eval') or d.get('extra',{}).get('objective_ms'), d['check']['ok'])
print({k:v for k,v in d.items() if 'objective' in k or 'launch' in k})"
