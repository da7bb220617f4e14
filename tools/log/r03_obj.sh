# the one-launch objective: one 8-wave workgroup per CU on 16-frame tiles against two 4-wave workgroups on 8-frame tiles
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for W in 8 4 4s1 4s2 4s3; do
  SPECINV_OBJ_STAGGER=${W#4s} SPECINV_OBJ_WAVES=${W%%s*} python bench.py --workload C5 --outer 5 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 waves $W', round(d['ms_per_step'],3), 'objective launch_ms', round(d['roofline']['launch_ms'],4), d['check']['ok'])"
done
done
