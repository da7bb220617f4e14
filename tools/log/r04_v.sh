cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python3 -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|^FAILED|Error" | head -5
done
