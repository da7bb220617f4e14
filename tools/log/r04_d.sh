cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r04_d_tests.txt
cat gpurun_out/r04_d_tests.txt
for p in frame fused fused_prespec; do python3 tools/near_zero_event.py 0.3 $p; done > gpurun_out/r04_near_zero_event.txt 2>&1
cat gpurun_out/r04_near_zero_event.txt
