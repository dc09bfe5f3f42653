#!/bin/bash
# A/B of one environment switch inside one gpurun call: tests, layer tables with the switch set / unset, three alternating bench pairs.
#   usage: r5_abenv.sh <outdir> <VAR=1> [config [precision]]        (AB_TESTS: test files to run first; default the 16-bit path's)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; V=$2; CFG=${3:-surreal-depth1}; PREC=${4:-bf16cl}; mkdir -p $O
T=${AB_TESTS:-tests/test_cl16_gpu.py tests/test_cl16_b100_gpu.py tests/test_fp16_gpu.py tests/test_cl16_oracle_gpu.py}
timeout -k 10 900 python3 -m pytest $T -m gpu -x -q > $O/tests.log 2>&1 || { tail -25 $O/tests.log; exit 1; }
tail -n 2 $O/tests.log
lt() { env $1 timeout -k 10 250 python3 tools/layer_table.py $CFG --precision $PREC --csv $O/layers_$2.csv > $O/layers_$2.txt 2>&1 || { tail -5 $O/layers_$2.txt; exit 1; }; echo "$2: $(tail -n 1 $O/layers_$2.txt)"; }
lt $V off && lt X=1 on || exit 1
B="--config $CFG --precision $PREC --steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary"
for v in "$V" "X=1" "$V" "X=1" "$V" "X=1"; do
  env $v timeout -k 10 200 python3 bench.py $B 2> $O/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), 'ms', round(d['value'],1))" || { tail -3 $O/bench.err; exit 1; }
done
