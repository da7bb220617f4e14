cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fast.py tests/test_gpu_bench_sizes.py tests/test_gpu_parity.py -q -x -n 4 -k "phase_init or c2 or c4" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pi_kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/pi_kt.log 2>&1
grep -h "phase_init_pairs\|fused_istft" gpurun_out/pi_kt/*/*kernel_stats.csv | cut -c1-160
for i in 1 2; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', round(d['ms_per_step'],3), d['check']['ok'], d['check']['reference']['max_abs_dsc_lin'])"; done
