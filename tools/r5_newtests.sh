#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_as_trainer_gpu.py tests/test_cl16_b100_gpu.py tests/test_f32x6_parity_gpu.py -m gpu -q > $O/newtests.log 2>&1; rc=$?
grep -E "^(FAILED|ERROR)|passed|failed" $O/newtests.log | tail -n 40
B="--steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary"
for v in "DCV_TORCH_GRAD_ADDS=1" "X=1" "DCV_TORCH_GRAD_ADDS=1" "X=1"; do
  env $v timeout -k 10 200 python3 bench.py $B 2> $O/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 $v', round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['config']['hip_launches_per_step'])" || { tail -3 $O/bench.err; exit 1; }
done
for v in "DCV_TORCH_GRAD_ADDS=1" "X=1" "DCV_TORCH_GRAD_ADDS=1" "X=1"; do
  env $v timeout -k 10 200 python3 bench.py --config surreal-depth1 --precision bf16cl $B 2> $O/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16cl $v', round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['config']['hip_launches_per_step'])" || { tail -3 $O/bench.err; exit 1; }
done
exit $rc
