cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_rtisi.py tests/test_gpu_stream.py tests/test_gpu_bench_sizes.py tests/test_gpu_random_configs.py tests/test_gpu_api.py tests/test_gpu_parity.py -q -m gpu -k "rtisi or stream or c3 or RTISI" 2>&1 | tail -40) > gpurun_out/r02_rtisi_tests.log 2>&1
tail -n 12 gpurun_out/r02_rtisi_tests.log
timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two waves:', d['ms_per_step'], d['roofline']['dependent_steps_per_s'], d['check'])"
timeout 300 python bench.py --workload C3 --asym --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two waves asym:', d['ms_per_step'], d['roofline']['dependent_steps_per_s'], d['check'])"
SPECINV_RTISI_TWO_WAVES=0 timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one wave:', d['ms_per_step'], d['roofline']['dependent_steps_per_s'], d['check'])"
