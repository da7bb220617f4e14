# round 3, first measurement pass: test-suite, every bench workload, rocprofv3 passes of C2 / C4 / C5 / C1
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -n 4 2>&1 | tail -80 > gpurun_out/r03_tests2.log
for W in C2 C4 C3 C5 C1; do
  python bench.py --workload $W --steps 5 --warmup 1 > gpurun_out/r03_bench_$W.json 2> gpurun_out/r03_bench_$W.err
done
python bench.py --workload C5 --c5-variant wolfe --outer 10 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_C5_wolfe.json 2> gpurun_out/r03_bench_C5_wolfe.err
python bench.py --workload C5 --c5-variant main --outer 5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_C5_main.json 2> gpurun_out/r03_bench_C5_main.err
SPECINV_EXACT=1 python bench.py --workload C2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_C2_exact.json 2> gpurun_out/r03_bench_C2_exact.err
bash tools/profile_workloads.sh r03 C2 C4 C3 C5
du -sh gpurun_out; rm -rf gpurun_out/*_kt/*/*.db gpurun_out/*/*/*_agent_info.csv
tail -5 gpurun_out/r03_tests2.log
for W in C2 C4 C3 C5 C1 C5_wolfe C5_main C2_exact; do cut -c1-400 gpurun_out/r03_bench_$W.json; tail -2 gpurun_out/r03_bench_$W.err; done
