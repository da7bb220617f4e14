# k_fused4<8, ADMM> at BASELINE C4: three waves per SIMD finish after 193 / 226 / 271 k ticks; chunk triples of unequal length
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for S in "0,0" "4,6" "2,3" "4,5"; do
  echo "skew $S: $(SPECINV_K4_SKEW=$S python bench.py --workload C4 --steps 5 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4))")"
done
done
