#!/bin/bash
# The ONE runner for GPU experiments: `gpurun --timeout S -- 'bash tools/gpu.sh <tag> <command ...>'`.
# Sets up the scratch directory the way rocprofv3 wants it, runs the command from the repository root, keeps its output as
# gpurun_out/<tag>.log.  What an experiment measured and decided goes into tools/log/EXPERIMENTS.md, not into a script per run.
tag=$1; shift
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
( eval "$@" ) > "gpurun_out/$tag.log" 2>&1
rc=$?
tail -n "${GPU_SH_TAIL:-60}" "gpurun_out/$tag.log"
exit $rc
