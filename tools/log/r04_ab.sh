cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|Error" | head -10 > gpurun_out/r04_gpu_tests_tail.txt
cat gpurun_out/r04_gpu_tests_tail.txt
python3 __graft_entry__.py --smoke 2>&1 | tail -1
for v in baseline wolfe; do
python3 bench.py --workload C5 --c5-variant $v --no-extra > gpurun_out/r04_bench_C5_$v.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('gpurun_out/r04_bench_C5_$v.json').read().strip().splitlines()[-1]); print('C5 $v', round(d['value']/1e6,2), round(d['ms_per_step'],2), d['check']['ok'], d['config']['lbfgs']['decisions'])"
done
