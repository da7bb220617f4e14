cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_lbfgs.py -q -m gpu -k "one_launch" 2>&1 | tail -30) > gpurun_out/r02_obj_tests.log 2>&1
tail -n 8 gpurun_out/r02_obj_tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_c5_kt -- python3 tools/bench_configs.py C5 > gpurun_out/r02_c5_kt.log 2>&1
tail -4 gpurun_out/r02_c5_kt.log
find gpurun_out/r02_c5_kt -name "*kernel_stats.csv" | head -1 | xargs head -12
