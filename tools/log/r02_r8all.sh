cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants/libspecinv_r8all.so
for cfg in "1024 128 1024 64 gla" "1024 512 1024 64 gla" "1024 128 1024 64 admm" "1024 512 1024 64 admm"; do
  set -- $cfg
  for lib in "" "$V"; do
    echo -n "n_fft $1 hop $2 T $3 B $4 $5 ${lib:+r8all}: "; SPECINV_LIB=$lib python tools/bench_iter.py --n-fft $1 --hop $2 --frames $3 --batch $4 --method $5 2>/dev/null | tail -1 | cut -c1-100
  done
done
