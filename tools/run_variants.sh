# run tools/bench_iter.py against every built variant (dev tool; on the GPU box)
for f in spectrogram_inversion_amd/variants/libspecinv_*.so; do
  echo "== $(basename $f)"
  SPECINV_LIB=$PWD/$f python3 tools/bench_iter.py ${BENCH_ARGS:-} 2>&1 | grep -E "chunk|rror"
done
