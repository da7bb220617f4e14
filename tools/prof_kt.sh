cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 tools/bench_iter.py --launches 20 --rounds 1 > gpurun_out/prof_kt.log 2>&1
ls -R gpurun_out/prof_kt | head -20
