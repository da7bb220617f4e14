cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants
python -m pytest tests/test_gpu_fast.py -m gpu -q -x -k "skewed or time_domain" 2>&1 | tail -2
for i in 1 2; do
for L in fair0 fair16; do
for S in 0 8 10 12 14; do echo "$L skew $S: $(SPECINV_LIB=$V/libspecinv_$L.so SPECINV_TD_SKEW=$S python tools/bench_iter.py --launches 100 --rounds 3 2>&1 | tail -1 | cut -c40-70)"; done
done
done
for S in 10 12; do
SPECINV_TD_SKEW=$S SPECINV_TD_STAMP_DUMP=gpurun_out/td_waves_$S.txt python tools/td_stamps.py 2>&1 | grep "per-wave" | tail -1
done
