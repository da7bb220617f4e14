# chunk skew with the roles taken from the dispatch order (4-wave workgroups, first half of the waves = older) against 8-wave
# workgroups where the role is the wave's index in the workgroup / 4
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "wgw 4 (default): $(python bench.py --workload C2 --steps 5 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4))")"
  for S in 8 10; do
  echo "wgw 8 skew $S: $(SPECINV_FUSED_WGW=8 SPECINV_TD_SKEW=$S python bench.py --workload C2 --steps 5 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4), d['config']['launch_geometry'])")"
  done
done
