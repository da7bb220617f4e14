# round 4, first call: the restructured bench.py (live PMC child, H2D legs, extra workloads) + the GPU suite on the r03 kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python3 bench.py --keep-pmc gpurun_out/r04_pmc_raw > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err ) 2> gpurun_out/r04_bench_default.time
tail -c 600 gpurun_out/r04_bench_default.err
cat gpurun_out/r04_bench_default.time
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r04_a_tests.txt
cat gpurun_out/r04_a_tests.txt
