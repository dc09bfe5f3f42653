#!/bin/bash
# A/B of the bench under an env toggle: tools/ab_bench.sh VAR
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "" "1"; do
  echo "=== $1=${v:-unset}"
  if [ -n "$v" ]; then export $1=1; else unset $1; fi
  timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-minimal 2>&1 | grep '^{"metric"' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['losses_last_step'])" || exit 1
done
