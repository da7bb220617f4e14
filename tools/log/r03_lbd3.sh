cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_lbfgs.py -q -x 2>&1 | tail -3
for d in 0 1; do
SPECINV_LBFGS_DEVICE=$d python bench.py --workload C5 --outer 10 --steps 3 --warmup 1 --no-cpu-baseline --no-check 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 device=$d', round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4))"
done
for d in 1; do
SPECINV_LBFGS_DEVICE=$d timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lbd_kt$d -- python3 bench.py --workload C5 --outer 2 --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/lbd_kt$d.log 2>&1
done
# and without the profiler, without the per-evaluation events
for d in 0 1; do
SPECINV_LBFGS_DEVICE=$d SPECINV_BENCH_NO_EVENTS=1 python bench.py --workload C5 --outer 10 --steps 3 --warmup 1 --no-cpu-baseline --no-check 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 no events device=$d', round(d['ms_per_step'],3))"
done
