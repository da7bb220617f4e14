# round 2, first GPU call: the new tests, the whole GPU suite, every bench workload, the IEEE study
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_bench_sizes.py tests/test_gpu_nccl.py -q -m gpu -x --durations=8 2>&1 | tail -40) > gpurun_out/r02_newtests.log 2>&1
(timeout 1800 python -m pytest tests -q -m gpu --durations=15 -x --deselect tests/test_gpu_bench_sizes.py --deselect tests/test_gpu_nccl.py 2>&1 | tail -60) > gpurun_out/r02_gputests.log 2>&1
for w in C2 C4 C3 C5 C1; do
  timeout 600 python bench.py --workload $w > gpurun_out/r02_bench_$w.json 2> gpurun_out/r02_bench_$w.err
  tail -c 1500 gpurun_out/r02_bench_$w.json
done
timeout 900 python tools/ieee_study.py > gpurun_out/r02_ieee_stdout.txt 2>&1
tail -5 gpurun_out/r02_newtests.log gpurun_out/r02_gputests.log
