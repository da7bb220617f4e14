# whole GPU suite on the new arithmetic, then C2 / C4 / C5 / C3 bench lines: committed library (head) against the new one, same box
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants
python -m pytest tests -m gpu -q -n 4 2>&1 | tail -40
for i in 1 2; do
for W in C2 C4 C5 C3; do
  echo "== $W head"; SPECINV_LIB=$V/libspecinv_head.so python bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('launch_ms'))"
  echo "== $W new"; python bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('launch_ms'))"
done
done
