# run tools/bench_iter.py against every built variant, interleaved over several rounds (dev tool; GPU box)
for round in 1 2 3; do
for f in spectrogram_inversion_amd/variants/libspecinv_*.so; do
  echo "== $(basename $f) round $round"
  SPECINV_LIB=$PWD/$f python3 tools/bench_iter.py ${BENCH_ARGS:-} 2>&1 | grep -E "chunk|rror"
done
done
