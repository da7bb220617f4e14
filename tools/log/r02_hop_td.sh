cd $GRAFT_REPO_ROOT
for cfg in "2048 333 1024 32" "2048 600 1024 32" "2048 768 1024 64" "2048 1000 1024 32" "2048 200 1024 32" "1024 160 2048 32" "1024 300 2048 32" "1024 400 2048 32" "1024 100 2048 32" "512 100 4096 32" "512 200 4096 32"; do
  set -- $cfg
  for k in "" "--keep-state"; do
    echo -n "n_fft $1 hop $2 T $3 B $4 $k: "; python tools/bench_iter.py --n-fft $1 --hop $2 --frames $3 --batch $4 $k 2>/dev/null | tail -1 | cut -c1-90
  done
done
