cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r04_h_tests.txt
cat gpurun_out/r04_h_tests.txt
SPECINV_GENERIC_INPLACE=1 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random_configs.py tests/test_gpu_properties.py tests/test_gpu_autograd.py -q -m gpu 2>&1 | tail -5 > gpurun_out/r04_h_tests_ip1.txt
cat gpurun_out/r04_h_tests_ip1.txt
