#!/bin/bash
# N alternating bench pairs (toggle unset / set) inside ONE gpurun call: usage  ab_bench.sh DCV_NO_XYZ [pairs] [extra bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
T=$1; P=${2:-4}; shift; shift
for i in $(seq 1 $P); do
a=$(timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 12 --warmup 3 "$@" 2>/dev/null | python3 -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
b=$(env $T=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 12 --warmup 3 "$@" 2>/dev/null | python3 -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
echo "pair $i: new $a ms   $T=1 $b ms"
done
