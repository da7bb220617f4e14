cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 1 0; do
rm -rf gpurun_out/r04_wolfe_kt$v
SPECINV_LBFGS_DEVICE_WOLFE=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04_wolfe_kt$v -- python3 bench.py --workload C5 --c5-variant wolfe --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-extra --no-pmc --no-h2d > gpurun_out/r04_wolfe_kt$v.log 2>&1
python3 - $v <<'PY'
import csv, glob, sys, collections
v=sys.argv[1]
f=sorted(glob.glob(f'gpurun_out/r04_wolfe_kt{v}/*/*kernel_trace.csv'))[-1]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//2:]
span=(int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp']))/1e6
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)/1e6
print('device' if v=='1' else 'host', 'kernels', len(rows), 'span ms', round(span,2), 'busy ms', round(busy,2))
d=collections.defaultdict(lambda:[0,0])
for r in rows:
    n=r['Kernel_Name'].split('(')[0][-45:]
    d[n][0]+=1; d[n][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for n,(c,t) in sorted(d.items(), key=lambda kv:-kv[1][1])[:12]: print('  %-46s %5d %9.1f us total %7.1f us each'%(n,c,t,t/c))
PY
done
