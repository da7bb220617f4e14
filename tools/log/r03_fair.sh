# the two waves of a SIMD: the older one wins every tie and finishes after 69 % of the launch; time-sliced priority, slice 2^K ticks
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants
export SPECINV_TD_SKEW=0
for i in 1 2; do
  for K in 0 11 12 13 14 15 16; do
    echo "== fair $K: $(SPECINV_LIB=$V/libspecinv_fair$K.so python tools/bench_iter.py --launches 100 --rounds 3 2>&1 | tail -1 | cut -c40-70)"
  done
done
python tools/td_stamps.py 2>&1 | grep "per-wave\|iteration"
