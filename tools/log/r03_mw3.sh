# late launches of k_fused4_td<16> at two waves per SIMD (shipped) against three (168 registers, 12-wave workgroups, 3072 wave slots)
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants
for i in 1 2; do
  python tools/bench_iter.py --launches 60 --rounds 3 2>&1 | tail -1
  SPECINV_LIB=$V/libspecinv_tdmw3.so SPECINV_FUSED_SLOTS=3072 SPECINV_FUSED_WGW=12 python tools/bench_iter.py --launches 60 --rounds 3 2>&1 | tail -1
  SPECINV_LIB=$V/libspecinv_tdmw3b.so SPECINV_FUSED_SLOTS=3072 SPECINV_FUSED_WGW=12 python tools/bench_iter.py --launches 60 --rounds 3 2>&1 | tail -1
done
