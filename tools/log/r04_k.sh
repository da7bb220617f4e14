# whole C2 step against the chunk skew, on the reference-order kernels (r03 tuned it on the approximate ones)
cd $GRAFT_REPO_ROOT
for rnd in 1 2; do for s in 6 8 10 12; do SPECINV_TD_SKEW=$s python3 bench.py --no-extra --no-pmc --no-h2d --no-cpu-baseline --no-check --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skew $s', round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],4))"; done; done
