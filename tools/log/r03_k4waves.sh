cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants
SPECINV_LIB=$V/libspecinv_k4stamps.so SPECINV_K4_STAMP_DUMP=gpurun_out/k4_waves.txt python bench.py --workload C4 --steps 1 --warmup 0 --no-cpu-baseline --no-check > /dev/null 2>&1
python - <<'PY'
import numpy as np, collections
rows=[tuple(int(v) for v in ln.split()) for ln in open('gpurun_out/k4_waves.txt')]
r=np.array(rows,dtype=np.int64)
wave,xcc,hw,beg,end,fr=r[:,1],r[:,2]&0xf,r[:,3],r[:,4],r[:,5],r[:,6]
wid=hw&0xf; dur=end-beg
print('waves',len(r),'wave slot ids',collections.Counter(wid.tolist()))
for s in sorted(set(wid.tolist())):
    m=wid==s
    print(f"slot {s}: waves {m.sum()} frames mean {fr[m].mean():.1f} duration mean {dur[m].mean():.0f} min {dur[m].min()} max {dur[m].max()}; wave-in-workgroup indices {sorted(set((wave[m]%12).tolist()))}")
print('overall mean',dur.mean(),'max',dur.max(), 'max/mean', dur.max()/dur.mean())
PY
