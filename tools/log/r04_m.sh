cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" > gpurun_out/r04_gpu_tests_tail.txt
cat gpurun_out/r04_gpu_tests_tail.txt
for w in 0 4; do
  SPECINV_STAMP_LIB=$PWD/spectrogram_inversion_amd/variants/libspecinv_stamps$w.so python3 tools/obj_stamps.py 2>&1 | grep -v "^$" | tail -16
done
