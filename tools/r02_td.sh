cd $GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/td_kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload C2 --steps 2 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls -t gpurun_out/td_kt/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-170
