# which of today's changes moved C3 (k_rtisi_fast, a lone wave per SIMD) and C1 (k_semi + k_ola): same box, variant libraries
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants
for i in 1 2; do
for L in head new nohalf nopk norsq; do
  if [ $L = new ]; then unset SPECINV_LIB; else export SPECINV_LIB=$V/libspecinv_$L.so; fi
  for W in C3 C1; do
    echo "$L $W $(python bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
done
