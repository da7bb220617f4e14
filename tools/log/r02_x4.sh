cd $GRAFT_REPO_ROOT
bash tools/r02_quick.sh C2 C4 C3 C5
SPECINV_LIB=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants/libspecinv_rtpk.so timeout 300 python bench.py --workload C3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 packed variant', round(d['ms_per_step'],3), d['check']['ok'])"
