cd $GRAFT_REPO_ROOT
[ -n "$WITH_SWEEP" ] && python3 tools/sweep.py --tag r05 --out gpurun_out/r05_sweep.json > gpurun_out/r05_sweep.log 2>&1; tail -2 gpurun_out/r05_sweep.log
for v in baseline wolfe main memory; do
python3 bench.py --workload C5 --c5-variant $v --no-extra --no-pmc > gpurun_out/r05_bench_C5_$v.json 2>/dev/null
done
SPECINV_OBJ_WALK=0 python3 bench.py --workload C5 --no-extra --no-pmc --no-cpu-baseline > gpurun_out/r05_bench_C5_tiles.json 2>/dev/null
python3 bench.py --workload C4 --no-extra > gpurun_out/r05_bench_C4.json 2>/dev/null
python3 bench.py --workload C3 --no-extra > gpurun_out/r05_bench_C3.json 2>/dev/null
python3 bench.py --workload C3 --asym --no-extra > gpurun_out/r05_bench_C3_asym.json 2>/dev/null
python3 bench.py --workload C1 --no-extra > gpurun_out/r05_bench_C1.json 2>/dev/null
SPECINV_EXACT=0 python3 bench.py --no-extra > gpurun_out/r05_bench_C2_approx.json 2>/dev/null
python3 tools/bench_generic_r05.py > gpurun_out/r05_generic.txt 2>&1
python3 tools/bench_generic_r04.py >> gpurun_out/r05_generic.txt 2>&1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    if 'value' not in d: continue
    r=d.get('roofline',{})
    print(f.split('bench_')[1][:-5], round(d['value']/1e6,2), round(d['ms_per_step'],3), r.get('limiter'), round(r.get('limiter_frac') or 0,3), round(r.get('frac') or 0,3), (d.get('check') or {}).get('ok'))
PY
