# round 4: the reference's operation chain (ref_rcp_abs2) - accuracy on the fixtures and cost against the default approximations
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 tools/refchain_study.py spectrogram_inversion_amd/variants/libspecinv_rc1b.so spectrogram_inversion_amd/variants/libspecinv_rc1d.so spectrogram_inversion_amd/variants/libspecinv_rc2d.so > gpurun_out/r04_refchain.log 2>&1
grep -E "strict gate:|ms per late|^==" gpurun_out/r04_refchain.log | tail -60
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fast.py tests/test_gpu_semi.py tests/test_gpu_two_process.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_b_tests.txt
cat gpurun_out/r04_b_tests.txt
