# the whole GPU suite (new files first)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_bench_sizes.py tests/test_gpu_nccl.py -q -m gpu --durations=8 2>&1 | tail -40) > gpurun_out/r02_newtests.log 2>&1
(timeout 2400 python -m pytest tests -q -m gpu --durations=15 --deselect tests/test_gpu_bench_sizes.py --deselect tests/test_gpu_nccl.py 2>&1 | tail -80) > gpurun_out/r02_gputests.log 2>&1
tail -n 5 gpurun_out/r02_newtests.log; tail -n 5 gpurun_out/r02_gputests.log
