# rocprofv3 kernel trace of the other BASELINE configurations (C1, C3, C4 shard, C5); summary under gpurun_out/cfg_kt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cfg_kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg_kt -- python3 tools/bench_configs.py > gpurun_out/cfg_kt.log 2>&1
grep "^C" gpurun_out/cfg_kt.log
f=$(find gpurun_out/cfg_kt -name "*kernel_stats.csv" | head -1)
cut -c1-150 $f | head -25
