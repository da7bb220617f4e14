cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -k "exact" 2>&1 | tail -30
# what the exact kernels cost on the headline
python tools/bench_iter.py --launches 60 --rounds 3 2>&1 | tail -1
SPECINV_EXACT=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 exact', round(d['ms_per_step'],3), d['roofline']['launch_ms'], d['check']['ok'], d['check'].get('reference',{}).get('max_abs_dsc_lin'))"
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 default', round(d['ms_per_step'],3), d['roofline']['launch_ms'], d['check']['ok'], d['check'].get('reference',{}).get('max_abs_dsc_lin'))"
