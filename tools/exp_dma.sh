#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for e in "" STG "" STG; do
  echo "=== variant ${e:-base}"
  if [ -n "$e" ]; then export DCV_LIB=$PWD/dcvgan_amd/exp_$e.so; else unset DCV_LIB; fi
  timeout -k 10 200 python tools/kbench.py "$1" 2>&1 | grep -v "amdgpu.ids\|^layer" || exit 1
done
