#!/bin/bash
# the whole GPU suite (one process), log in gpurun_out/<dir>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-tests}; mkdir -p $O
timeout -k 10 ${2:-1000} python3 -m pytest tests -m gpu -x -q ${3:-} > $O/gpu_tests.log 2>&1; rc=$?
tail -n 15 $O/gpu_tests.log
exit $rc
