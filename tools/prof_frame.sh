# rocprofv3 kernel trace of the any-hop shapes (tools/bench_semi.py: k_hop + k_hop_tails) and of the L-BFGS direction
# passes (tools/bench_lbfgs_dir.py); summaries under gpurun_out/frame_kt, gpurun_out/lbdir_kt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/frame_kt gpurun_out/lbdir_kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/frame_kt -- python3 tools/bench_semi.py > gpurun_out/frame_kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lbdir_kt -- python3 tools/bench_lbfgs_dir.py > gpurun_out/lbdir_kt.log 2>&1
grep "path=" gpurun_out/frame_kt.log; grep "m=" gpurun_out/lbdir_kt.log
for d in frame_kt lbdir_kt; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); cut -c1-140 $f | head -8; done
