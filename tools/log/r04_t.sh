cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04_C1_kt -- python3 bench.py --workload C1 --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extra --no-pmc --no-h2d > gpurun_out/r04_C1_kt.log 2>&1
python3 - <<'PY'
import csv, glob
f=sorted(glob.glob('gpurun_out/r04_C1_kt/*/*kernel_trace.csv'))[-1]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//2:]
prev=None; gaps=[]; durs={}
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp']); n=r['Kernel_Name'][:40]
    durs.setdefault(n,[]).append(e-s)
    if prev is not None: gaps.append(s-prev)
    prev=e
for n,v in durs.items(): print(n, len(v), sum(v)/len(v)/1e3, 'us')
gaps=[g for g in gaps if g<50000]
print('gaps', len(gaps), sum(gaps)/len(gaps)/1e3, 'us mean', sorted(gaps)[len(gaps)//2]/1e3, 'median')
PY
