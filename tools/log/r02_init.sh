cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests/test_gpu_fast.py tests/test_gpu_parity.py tests/test_gpu_bench_sizes.py -q -m gpu 2>&1 | grep -E "^E|passed|failed" | head -20)
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2:', d['ms_per_step'], d['value'], d['roofline']['launch_ms'], d['check'])"
SPECINV_DISABLE_INIT_PAIRS=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 three-pass init:', d['ms_per_step'], d['value'], d['roofline']['launch_ms'])"
timeout 300 python bench.py --workload C4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4:', d['ms_per_step'], d['value'], d['roofline']['launch_ms'], d['check']['ok'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c_kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/r02c_kt.log 2>&1
find gpurun_out/r02c_kt -name "*kernel_stats.csv" | xargs ls -t | head -1 | xargs head -9 | cut -c1-160
