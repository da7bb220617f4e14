cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_lbfgs.py -q -m gpu 2>&1 | tail -5)
timeout 300 python tools/bench_configs.py C5 2>&1 | grep "C5"
