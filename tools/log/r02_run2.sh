cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -40) > gpurun_out/r02_gputests.log 2>&1
tail -n 4 gpurun_out/r02_gputests.log
for v in plate1 plate2 mw3; do echo "== variant $v"; SPECINV_LIB=$PWD/spectrogram_inversion_amd/variants/libspecinv_$v.so timeout 200 python3 tools/bench_iter.py 2>&1 | grep chunk; echo "== default"; timeout 200 python3 tools/bench_iter.py 2>&1 | grep chunk; done > gpurun_out/r02_variants.txt 2>&1
cat gpurun_out/r02_variants.txt
bash tools/profile_workloads.sh r02 C2 C4 C3 C5
