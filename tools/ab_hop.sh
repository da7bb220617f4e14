# k_hop (forced with SPECINV_SMALL_FRAMES=0) against k_semi + k_ola over the batch size (dev tool)
for cfg in "1 512" "2 1024" "4 1024" "8 1024" "16 1024" "32 1024"; do
  set -- $cfg
  a=$(SPECINV_SMALL_FRAMES=0 python tools/bench_iter.py --n-fft 1024 --hop 200 --batch $1 --frames $2 --launches 30 2>&1 | tail -1 | awk '{print $3}')
  b=$(SPECINV_DISABLE_HOP=1 python tools/bench_iter.py --n-fft 1024 --hop 200 --batch $1 --frames $2 --launches 30 2>&1 | tail -1 | awk '{print $3}')
  c=$(SPECINV_SMALL_FRAMES=0 python tools/bench_iter.py --n-fft 2048 --hop 441 --batch $1 --frames $2 --launches 30 2>&1 | tail -1 | awk '{print $3}')
  d=$(SPECINV_DISABLE_HOP=1 python tools/bench_iter.py --n-fft 2048 --hop 441 --batch $1 --frames $2 --launches 30 2>&1 | tail -1 | awk '{print $3}')
  echo "B=$1 T=$2  1024/200: hop $a semi $b   2048/441: hop $c semi $d"
done
