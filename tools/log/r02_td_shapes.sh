cd $GRAFT_REPO_ROOT
for cfg in "1024 128 512 64" "2048 256 512 64" "2048 1024 512 64" "1024 512 1024 64" "4096 1024 512 64" "512 128 2048 64" "1024 256 2048 32"; do
  set -- $cfg
  for k in "" "--keep-state"; do
    echo -n "n_fft $1 hop $2 T $3 B $4 $k: "; python tools/bench_iter.py --n-fft $1 --hop $2 --frames $3 --batch $4 $k 2>/dev/null | tail -1
  done
done
