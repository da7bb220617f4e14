cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_semi.py tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_random_configs.py tests/test_gpu_autograd.py tests/test_gpu_two_process.py -q -m gpu -x 2>&1 | tail -12 | cut -c1-300 > gpurun_out/r04_i_tests.txt
cat gpurun_out/r04_i_tests.txt
for g in 0 1; do SPECINV_SEMI_GATHER=$g python3 bench.py --workload C1 --steps 200 --warmup 10 --no-pmc --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C1 gather=$g', round(d['value']/1e6,2),'M', round(d['ms_per_step'],4),'ms', round(d['roofline']['launch_ms']*1e3,2),'us/iter', d['check']['ok'])"; done
