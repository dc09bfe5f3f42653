#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for e in 0 1 2 3; do
  echo "=== DCV_EXP=$e"; DCV_EXP=$e timeout -k 10 200 python tools/kbench.py "$1" 2>&1 | grep -v "amdgpu.ids\|^layer" || exit 1
done
