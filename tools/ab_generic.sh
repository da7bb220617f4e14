# bench_generic / bench_configs under two library variants (dev tool): ab_generic.sh <variantA> <variantB>
for v in $1 $2 $1 $2; do
echo "== $v"
SPECINV_LIB=$PWD/spectrogram_inversion_amd/variants/libspecinv_$v.so python3 tools/bench_generic.py 2>&1 | grep n_fft
SPECINV_LIB=$PWD/spectrogram_inversion_amd/variants/libspecinv_$v.so python3 tools/bench_configs.py 2>&1 | grep -E "^C[1345]"
done
