cd $GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests/test_gpu_fast.py tests/test_gpu_parity.py tests/test_gpu_bench_sizes.py -q -m gpu -x 2>&1 | grep -E "^E  |passed|failed|FAILED" | cut -c1-250 | head -20)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/td_kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload C2 --steps 2 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls -t gpurun_out/td_kt/*/*kernel_stats.csv | head -1); head -7 $f | cut -c1-150
python bench.py --workload C2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', round(d['ms_per_step'],3), d['check']['ok'])"
