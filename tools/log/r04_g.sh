cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in 0 default; do
  echo "== SPECINV_GENERIC_INPLACE=$m"
  if [ $m = default ]; then python3 tools/bench_generic_r04.py 2>&1; else SPECINV_GENERIC_INPLACE=$m python3 tools/bench_generic_r04.py 2>&1; fi
done | grep -v amdgpu.ids | tee gpurun_out/r04_generic_inplace.txt
python3 -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r04_g_tests.txt
cat gpurun_out/r04_g_tests.txt
SPECINV_GENERIC_INPLACE=1 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random_configs.py tests/test_gpu_properties.py tests/test_gpu_autograd.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r04_g_tests_ip1.txt
cat gpurun_out/r04_g_tests_ip1.txt
