# whole C2 step against the chunk skew (the evaluating launches and the initial ISTFT like less of it than the plain launches)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for S in 0 6 8 10 12; do
  echo "skew $S: $(SPECINV_TD_SKEW=$S python bench.py --workload C2 --steps 5 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4))")"
done
done
