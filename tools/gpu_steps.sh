#!/bin/bash
# Run a list of GPU steps one after another on the box; a step that FAILS (assertion, rc 1) does not stop the list, a step that is
# killed / times out (rc 124, 137, >= 128) does: nothing else is started on a GPU that may be in a bad state.
# usage: tools/gpu_steps.sh OUTDIR  "name|timeout_s|command" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$1; shift; mkdir -p "$O"
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; t=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${t}s)"; s=$(date +%s)
  timeout -k 10 "$t" bash -c "$cmd" > "$O/$name.log" 2> "$O/$name.err"; rc=$?
  echo "   rc $rc, $(( $(date +%s) - s )) s"; tail -n 3 "$O/$name.log" | cut -c1-300
  if [ $rc -ne 0 ]; then tail -n 5 "$O/$name.err" | cut -c1-300; fi
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "   killed / timed out: stopping here"; exit $rc; fi
done
exit 0
