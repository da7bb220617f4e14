cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_lbfgs.py -q -x -k "wolfe" 2>&1 | tail -30
