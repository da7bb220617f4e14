cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c1_kt -- python3 bench.py --workload C1 --steps 20 --warmup 2 --no-cpu-baseline --no-check > gpurun_out/c1_kt.log 2>&1
head -8 gpurun_out/c1_kt/*/*kernel_stats.csv | cut -c1-200
