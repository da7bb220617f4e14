# the chunk skew at the other overlaps of n_fft 2048 (same kernel body, 2048 waves): hop 256 (B 64, T 1024) and hop 1024
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for H in 256 1024; do
for S in 0 4 8 12; do
  echo "hop $H skew $S: $(SPECINV_TD_SKEW=$S python tools/bench_iter.py --hop $H --batch 64 --frames 1024 --launches 60 --rounds 3 2>&1 | tail -1 | cut -c1-80)"
done
done
done
