cd $GRAFT_REPO_ROOT
python tools/dbg_lbd.py 2>&1 | tail -40
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SPECINV_LBFGS_DEVICE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lbd_kt -- python3 bench.py --workload C5 --outer 2 --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/lbd_kt.log 2>&1
tail -2 gpurun_out/lbd_kt.log | cut -c1-300
