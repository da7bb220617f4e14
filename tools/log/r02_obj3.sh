cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_lbfgs.py tests/test_abi.py -q -m "gpu or not gpu" 2>&1 | tail -30) > gpurun_out/r02_obj_tests.log 2>&1
tail -n 6 gpurun_out/r02_obj_tests.log
timeout 300 python tools/bench_configs.py C5 2>&1 | grep "C5"
SPECINV_LBFGS_PACKED=0 timeout 300 python tools/bench_configs.py C5 2>&1 | grep "L_BFGS"
timeout 300 python bench.py --workload C5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
