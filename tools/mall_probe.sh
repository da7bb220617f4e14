for b in 4 8 16 32 64; do
for v in base nt0; do
echo "== B=$b $v"
SPECINV_LIB=$PWD/spectrogram_inversion_amd/variants/libspecinv_$v.so python3 tools/bench_iter.py --batch $b --chunks 8,16 --launches 100 2>&1 | grep -E "chunk|rror"
done
done
