#!/bin/bash
# A/B of one environment toggle that switches a NEW behaviour ON: usage  ab_toggle.sh DCV_XYZ [pairs] [extra bench args...]
# correctness first (the bitwise-reproducibility + training-step tests with the toggle on), then alternating bench pairs in one call.
cd "$GRAFT_REPO_ROOT" || exit 1
T=$1; N=${2:-3}; shift; shift
O=gpurun_out/ab_$T; mkdir -p $O
env $T=1 timeout -k 10 600 python3 -m pytest tests/test_fullwidth_gpu.py -k "training_step" -x -q -m gpu > $O/tests.log 2>&1 || { tail -n 30 $O/tests.log; exit 1; }
tail -n 1 $O/tests.log
for i in $(seq $N); do
a=$(timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary --steps 10 --warmup 3 "$@" 2>/dev/null | python3 -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
b=$(env $T=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary --steps 10 --warmup 3 "$@" 2>/dev/null | python3 -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
echo "base $a | $T=1 $b" | tee -a $O/pairs.txt
done
