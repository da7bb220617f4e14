cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | head -10 > gpurun_out/r04_gpu_tests_tail.txt
cat gpurun_out/r04_gpu_tests_tail.txt
python3 __graft_entry__.py --smoke 2>&1 | tail -1
( time python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err ) 2>&1 | grep real
for v in baseline wolfe main memory; do
python3 bench.py --workload C5 --c5-variant $v --no-extra > gpurun_out/r04_bench_C5_$v.json 2>/dev/null
done
python3 bench.py --workload C4 --no-extra > gpurun_out/r04_bench_C4.json 2>/dev/null
python3 bench.py --workload C3 --no-extra > gpurun_out/r04_bench_C3.json 2>/dev/null
python3 bench.py --workload C3 --asym --no-extra > gpurun_out/r04_bench_C3_asym.json 2>/dev/null
python3 bench.py --workload C1 --no-extra > gpurun_out/r04_bench_C1.json 2>/dev/null
SPECINV_EXACT=0 python3 bench.py --no-extra > gpurun_out/r04_bench_C2_approx.json 2>/dev/null
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    r=d.get('roofline',{})
    print(f.split('bench_')[1][:-5], round(d['value']/1e6,2), round(d['ms_per_step'],3), r.get('bound'), round(r.get('frac') or 0,3), r.get('traffic_source','')[:30], (d.get('check') or {}).get('ok'), round(d.get('value_incl_h2d',0)/1e6,1))
    for k,v in (d.get('extra',{}).get('workloads') or {}).items(): print('   ', k, round(v.get('value',0)/1e6,2), round(v.get('ms_per_step',0),3), round(v.get('frac') or 0,3), (v.get('check') or {}).get('ok'), v.get('error'))
PY
