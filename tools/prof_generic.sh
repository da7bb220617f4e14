# kernel-level breakdown of the generic (off the fused path) iteration; outputs under gpurun_out/prof_gen
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gen
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gen -- python3 tools/bench_generic.py > gpurun_out/prof_gen.log 2>&1
f=$(find gpurun_out/prof_gen -name "*kernel_stats.csv" | head -1)
cut -c1-160 $f | head -12
