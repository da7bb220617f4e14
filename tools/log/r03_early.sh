cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -n 4 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ek_kt -- python3 bench.py --workload C2 --steps 3 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2>&1
head -4 gpurun_out/ek_kt/*/*kernel_stats.csv | cut -c1-130
rm -rf gpurun_out/ek_kt
for i in 1 2 3; do python bench.py --workload C2 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['check']['ok'], d['check']['reference']['max_abs_dsc_lin'])"; done
