# Griffin-Lim at n_fft 1024 (k_fused4_td<8>, three waves per SIMD) at the C4 shard's size: chunk triples
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for S in "0,0" "3,4" "4,6" "5,8" "6,10"; do
  echo "skew $S: $(SPECINV_K4_SKEW=$S python tools/bench_iter.py --n-fft 1024 --frames 2048 --batch 32 --launches 100 --rounds 3 2>&1 | tail -1 | cut -c1-90)"
done
done
