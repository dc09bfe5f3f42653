#!/bin/bash
# A/B of kbench under an env toggle: tools/ab_kbench.sh VAR [filter]
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "" "1"; do
  echo "=== $1=${v:-unset}"
  if [ -n "$v" ]; then export $1=1; else unset $1; fi
  timeout -k 10 300 python tools/kbench.py $2 2>&1 | grep -v "amdgpu.ids\|^layer" || exit 1
done
