cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests/test_gpu_semi.py -q -m gpu -k one_launch 2>&1 | grep -E "^E|passed|failed|assert" | head -40)
timeout 300 python bench.py --workload C1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C1 gather:', d['ms_per_step'], d['value'], d['roofline']['launch_ms'], d['check'])"
SPECINV_DISABLE_GATHER=1 timeout 300 python bench.py --workload C1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C1 two launches:', d['ms_per_step'], d['value'], d['roofline']['launch_ms'])"
