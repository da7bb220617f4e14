cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 tools/bench_generic_r04.py 2>&1 | tee gpurun_out/r04_generic_before.txt
rm -rf gpurun_out/prof_gen
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gen -- python3 tools/bench_generic_r04.py > gpurun_out/prof_gen.log 2>&1
f=$(find gpurun_out/prof_gen -name "*kernel_stats.csv" | head -1)
cut -c1-200 $f | head -14
