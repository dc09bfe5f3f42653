#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5m; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_cl16_b100_gpu.py tests/test_cl16_gpu.py tests/test_grad_accumulation_gpu.py -m gpu -q > $O/newtests.log 2>&1; rc=$?
grep -E "^(FAILED|ERROR)|passed|failed|AssertionError: \(" $O/newtests.log | tail -n 40
B="--steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary"
for v in "DCV_NO_GATED_DGRAD=1" "X=1" "DCV_NO_GATED_DGRAD=1" "X=1"; do
  env $v timeout -k 10 200 python3 bench.py --config surreal-depth1 --precision bf16cl $B 2> $O/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16cl $v', round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['config']['hip_launches_per_step'])" || { tail -3 $O/bench.err; exit 1; }
done
exit $rc
