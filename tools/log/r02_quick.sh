# quick regression + timing pass: GPU suite, then every bench workload without the CPU baseline
cd $GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "^E  |passed|failed|FAILED" | cut -c1-250 | head -20)
for w in ${@:-C2 C4 C3 C5 C1}; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > gpurun_out/t_bench_$w.json 2> gpurun_out/t_bench_$w.err
  python -c "
import json
d=json.loads(open('gpurun_out/t_bench_$w.json').read().strip().splitlines()[-1])
print('$w', round(d['ms_per_step'],3), '%.4g'%d['value'], 'launch_ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'check', d['check']['ok'])"
done
