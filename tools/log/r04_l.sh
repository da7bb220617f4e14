cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/r04_gpu_tests_tail.txt
cat gpurun_out/r04_gpu_tests_tail.txt
python3 __graft_entry__.py --smoke 2>&1 | tail -1
( time python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err ) 2>&1 | grep real
python3 -c "
import json
d=json.loads(open('gpurun_out/r04_bench_default.json').read().strip().splitlines()[-1])
print(round(d['value']/1e6,1), round(d['ms_per_step'],3), d['roofline']['bound'], round(d['roofline']['frac'],3), d['roofline']['traffic_source'][:40], d['check']['ok'], round(d['value_incl_h2d']/1e6,1), round(d['bench_seconds']))
for k,v in d['extra']['workloads'].items(): print(' ', k, round(v.get('value',0)/1e6,2), round(v.get('ms_per_step',0),3), v.get('roofline',{}).get('bound'), round(v.get('frac') or 0,3), (v.get('check') or {}).get('ok'), v.get('error'))
"
