cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants
export SPECINV_LIB=$V/libspecinv_fair0.so
for S in 0 3 5; do echo "fair0 skew $S: $(SPECINV_TD_SKEW=$S python tools/bench_iter.py --launches 60 --rounds 3 2>&1 | tail -1)"; done
unset SPECINV_LIB
for S in 0 4; do
  SPECINV_TD_SKEW=$S SPECINV_TD_STAMP_DUMP=gpurun_out/td_waves_$S.txt python tools/td_stamps.py 2>&1 | grep "per-wave" | tail -1
done
python - <<'PY'
import numpy as np, collections
for S in (0,4):
    rows=[tuple(int(v) for v in ln.split()) for ln in open(f'gpurun_out/td_waves_{S}.txt')]
    r=np.array([x for x in rows if x[0]==40],dtype=np.int64)
    wave,hw,beg,end,fr=r[:,1],r[:,3],r[:,4],r[:,5],r[:,6]
    wid=hw&0xf; dur=end-beg
    for s in (0,1):
        m=wid==s
        print(f"skew {S} slot {s}: waves {m.sum()} frames mean {fr[m].mean():.1f} duration mean {dur[m].mean():.0f} min {dur[m].min()} max {dur[m].max()}; first-half waves in this slot: {(wave[m]<1024).mean():.2f}")
PY
