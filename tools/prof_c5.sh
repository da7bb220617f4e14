# kernel breakdown of the C5 (log-mel L_BFGS) evaluation; output under gpurun_out/prof_c5
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- python3 tools/bench_configs.py C5 > gpurun_out/prof_c5.log 2>&1
f=$(find gpurun_out/prof_c5 -name "*kernel_stats.csv" | head -1)
cut -c1-150 $f | head -16
tail -3 gpurun_out/prof_c5.log
