#!/bin/bash
# three builds of the library in one gpurun call (DCV_LIB_PATH): tests on each, layer table of the patch-staged layers, alternating bench triples
#   usage: r5_ab3.sh <outdir> <lib a> <lib b> <lib c>      (paths relative to the repo root)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
for L in "$@"; do
  n=$(basename $L .so)
  DCV_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout -k 10 600 python3 -m pytest tests/test_cl16_gpu.py tests/test_cl16_b100_gpu.py -m gpu -x -q > $O/tests_$n.log 2>&1 || { tail -15 $O/tests_$n.log; exit 1; }
  echo "$n: $(tail -n 1 $O/tests_$n.log)"
  DCV_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout -k 10 250 python3 tools/layer_table.py surreal-depth1 --precision bf16cl --csv $O/layers_$n.csv > $O/layers_$n.txt 2>&1 || { tail -5 $O/layers_$n.txt; exit 1; }
  echo "$n: $(tail -n 1 $O/layers_$n.txt)"
done
B="--config surreal-depth1 --precision bf16cl --steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary"
for r in 1 2 3; do for L in "$@"; do
  DCV_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout -k 10 200 python3 bench.py $B 2> $O/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $L .so)', round(d['ms_per_step'],2), 'ms', round(d['value'],1))" || { tail -3 $O/bench.err; exit 1; }
done; done
