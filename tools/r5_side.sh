#!/bin/bash
# weight gradients on the companion stream (ops_cl._wgrad_on_side): the 16-bit path's tests in the shipped mode, then alternating bench triples: in-stream / shipped; usage: r5_side.sh [config [precision]]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5ak; mkdir -p $O
T="tests/test_cl16_gpu.py tests/test_cl16_b100_gpu.py tests/test_cl16_oracle_gpu.py tests/test_fp16_gpu.py tests/test_grad_accumulation_gpu.py tests/test_side_streams_gpu.py tests/test_as_trainer_gpu.py"
timeout -k 10 900 python3 -m pytest $T -m gpu -x -q > $O/tests_on.log 2>&1 || { tail -25 $O/tests_on.log; exit 1; }; echo "shipped (the main chain's weight gradients on a companion stream): $(tail -n 1 $O/tests_on.log)"
B="--config ${1:-surreal-depth1} --precision ${2:-bf16cl} --steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary"
for r in 1 2 3; do for v in DCV_NO_WGRAD_SIDE=1 X=1; do
  env $v timeout -k 10 200 python3 bench.py $B 2> $O/bench.err | V="$v" python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['V'][:44], round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['losses_last_step']['loss_gen'])" || { tail -3 $O/bench.err; exit 1; }
done; done
