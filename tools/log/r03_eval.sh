# evaluating iterations of the signal-form Griffin-Lim: a launch of its own (k_eval_td) against the fused evaluating variants
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fast.py tests/test_gpu_bench_sizes.py -q -x -n 4 -k "gla or td or time_domain or c2 or wellcond or trace" 2>&1 | tail -4
for i in 1 2; do
for f in 0 1; do
SPECINV_TD_FUSED_EVAL=$f python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 fused_eval=$f', round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4), d['check']['ok'], d['check'].get('reference',{}).get('max_abs_dsc_lin'))"
done
done
