# round 4: the reference's operation order as the default arithmetic - whole GPU suite + the bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r04_c_tests.txt
cat gpurun_out/r04_c_tests.txt
python3 bench.py > gpurun_out/r04_bench_refchain.json 2> gpurun_out/r04_bench_refchain.err
tail -c 300 gpurun_out/r04_bench_refchain.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r04_bench_refchain.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'], d['check']['ok'], d['check']['reference'])
for k,v in d['extra']['workloads'].items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('frac'), v.get('error'))
"
