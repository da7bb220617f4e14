# late launches of k_fused4_td<16> at two waves per SIMD (shipped) against three (168 registers, no spills now: the sample window is
# not carried, twiddles from LDS), with and without chunk triples
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants
for i in 1 2; do
  echo "2 waves: $(python tools/bench_iter.py --launches 100 --rounds 3 2>&1 | tail -1 | cut -c40-75)"
  for S in "0,0" "2,3" "4,6"; do
    echo "3 waves skew $S: $(SPECINV_LIB=$V/libspecinv_tdmw3.so SPECINV_FUSED_SLOTS=3072 SPECINV_FUSED_WGW=12 SPECINV_K4_SKEW=$S python tools/bench_iter.py --launches 100 --rounds 3 2>&1 | tail -1 | cut -c40-75)"
  done
done
