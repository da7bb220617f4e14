cd $GRAFT_REPO_ROOT
for w in C3 C2 C5; do
for lib in libspecinv.so variants/libspecinv_nopk.so; do
SPECINV_LIB=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/$lib timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $lib', round(d['ms_per_step'],3), d['check']['ok'])"
done
done
