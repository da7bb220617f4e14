cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_lbfgs.py -q -m gpu 2>&1 | tail -5)
timeout 300 python bench.py --workload C5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['launch_ms'], d['roofline']['frac'], d['check']['ok'])"
timeout 300 python tools/bench_configs.py C5 2>&1 | grep "C5"
