cd $GRAFT_REPO_ROOT
for i in 1 2; do
python bench.py --workload C2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 shipped', round(d['ms_per_step'],3), d['roofline']['launch_ms'], d['check']['ok'])"
SPECINV_LIB=$GRAFT_REPO_ROOT/spectrogram_inversion_amd/variants/libspecinv_tdmw3.so SPECINV_FUSED_SLOTS=3072 SPECINV_FUSED_WGW=12 python bench.py --workload C2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 mw3    ', round(d['ms_per_step'],3), d['roofline']['launch_ms'], d['check']['ok'], d['config']['launch_geometry'])"
done
