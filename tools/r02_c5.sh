cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --workload C5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['launch_ms'], d['roofline']['frac'], d['check'])"
timeout 300 python bench.py --workload C5 --no-cpu-baseline --outer 6 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['launch_ms'], d['roofline']['evaluations_timed'])"
