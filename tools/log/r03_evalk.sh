# the evaluation of x_t as a kernel of its own after the plain iteration kernel (same stream) against the fused evaluating variant
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in default ev23 ev_lds; do
  if [ $L = default ]; then unset SPECINV_LIB; else export SPECINV_LIB=$V/libspecinv_$L.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ek_kt -- python3 bench.py --workload C2 --steps 2 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2>&1
  echo "== $L: $(grep -h k_eval_td gpurun_out/ek_kt/*/*kernel_stats.csv | cut -c1-110)"
  rm -rf gpurun_out/ek_kt
done
unset SPECINV_LIB
for i in 1 2 3; do
  for S in 0 1; do
    echo "== SPECINV_EVAL_KERNEL=$S $(SPECINV_EVAL_KERNEL=$S python bench.py --workload C2 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['check']['ok'], d['check']['reference']['max_abs_dsc_lin'])")"
  done
done
