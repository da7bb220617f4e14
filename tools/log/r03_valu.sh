# late launch of k_fused4_td<16>: constant twiddles as packed products with a scalar-register operand (PKCONST), rsq projection
# with packed factors (TD_RSQ) - against the committed library (head) and the two switches off (base)
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants
for i in 1 2 3; do
  for L in head base pk; do
    echo "== $L"; SPECINV_LIB=$V/libspecinv_$L.so python tools/bench_iter.py --launches 60 --rounds 3 2>&1 | tail -1
  done
  echo "== new"; python tools/bench_iter.py --launches 60 --rounds 3 2>&1 | tail -1
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fast.py tests/test_gpu_bench_sizes.py -m gpu -q -x -n 4 2>&1 | tail -15
python bench.py --workload C2 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | cut -c1-600
