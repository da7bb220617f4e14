# the bench lines of the committed tree (after profiles/traffic.json was refreshed on the same sources): traffic_stale false
cd $GRAFT_REPO_ROOT
for W in C2 C4 C3 C5 C1; do
  python bench.py --workload $W --steps 5 --warmup 1 > gpurun_out/r03_bench_$W.json 2> gpurun_out/r03_bench_$W.err
done
python bench.py --workload C5 --c5-variant wolfe --outer 10 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_C5_wolfe.json 2>/dev/null
python bench.py --workload C5 --c5-variant main --outer 5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_C5_main.json 2>/dev/null
python bench.py --workload C5 --c5-variant memory --outer 8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_C5_memory.json 2>/dev/null
SPECINV_EXACT=1 python bench.py --workload C2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_C2_exact.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 2 > gpurun_out/r03_bench_C2_driver_cmd.json 2>/dev/null
for f in gpurun_out/r03_bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],3), round(d['value']/1e6,2), d['roofline'].get('traffic_stale'), d['check'].get('ok'))
PY
done
