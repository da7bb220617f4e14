# round-2 closing run: whole GPU suite, smoke, every bench workload, profiles of every workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -30) > gpurun_out/r02_gputests.log 2>&1
tail -n 3 gpurun_out/r02_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for w in C2 C4 C3 C5 C1; do
  timeout 600 python bench.py --workload $w > gpurun_out/r02_bench_$w.json 2> gpurun_out/r02_bench_$w.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r02_bench_$w.json').read().strip().splitlines()[-1])
print('$w', round(d['ms_per_step'],3), '%.4g'%d['value'], d['unit'], 'launch_ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'check', d['check']['ok'], 'cpu', '%.3g'%d['cpu_baseline']['value'])"
done
timeout 300 python bench.py --workload C3 --asym --no-cpu-baseline > gpurun_out/r02_bench_C3_asym.json 2>/dev/null
bash tools/profile_workloads.sh r02 C2 C4 C3 C5 > /dev/null 2>&1
ls gpurun_out | grep -c r02_
