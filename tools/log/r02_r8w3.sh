cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants/libspecinv_r8w3.so
for i in 1 2; do
for lib in "" "$V"; do
  SPECINV_LIB=$lib python bench.py --workload C4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4', '${lib:+r8w3}', round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4), d['check']['ok'], d['config']['launch_geometry'])"
done
done
for lib in "" "$V"; do
  echo -n "GLA 1024/256 B32 T2048 ${lib:+r8w3}: "; SPECINV_LIB=$lib python tools/bench_iter.py --n-fft 1024 --hop 256 --frames 2048 --batch 32 2>/dev/null | tail -1
  echo -n "GLA 1024/256 B96 T1024 ${lib:+r8w3}: "; SPECINV_LIB=$lib python tools/bench_iter.py --n-fft 1024 --hop 256 --frames 1024 --batch 96 2>/dev/null | tail -1
done
