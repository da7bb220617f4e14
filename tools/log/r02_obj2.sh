cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_lbfgs.py -q -m gpu -k "one_launch" 2>&1 | tail -30) > gpurun_out/r02_obj_tests.log 2>&1
tail -n 4 gpurun_out/r02_obj_tests.log
timeout 200 python tools/obj_stamps.py 2>&1 | tail -20
timeout 300 python tools/bench_configs.py C5 2>&1 | grep "C5 log-mel"
