#!/bin/bash
# A/B of two builds of the library inside one gpurun call: the shipped one and dcvgan_amd/alt_libdcvgan_hip.so (built here with EXTRA_HIPCC_FLAGS, selected
# through DCV_LIB_PATH).   usage: r5_ablib.sh <outdir> <config> <precision> [layer filter]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; CFG=${2:-surreal-depth1}; PREC=${3:-bf16cl}; FILT=${4:-}; mkdir -p $O
ALT=$GRAFT_REPO_ROOT/dcvgan_amd/alt_libdcvgan_hip.so
if [ -n "$AB_TESTS" ]; then timeout -k 10 900 python3 -m pytest $AB_TESTS -m gpu -x -q > $O/tests.log 2>&1 || { tail -25 $O/tests.log; exit 1; }; tail -n 2 $O/tests.log; fi
lt() { env $1 timeout -k 10 300 python3 tools/layer_table.py $CFG --precision $PREC --filter "$FILT" --csv $O/layers_$2.csv > $O/layers_$2.txt 2>&1 || { tail -5 $O/layers_$2.txt; exit 1; }; echo "$2: $(tail -n 1 $O/layers_$2.txt)"; }
lt DCV_LIB_PATH=$ALT alt && lt X=1 base || exit 1
B="--config $CFG --precision $PREC --steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary"
for v in "DCV_LIB_PATH=$ALT" "X=1" "DCV_LIB_PATH=$ALT" "X=1" "DCV_LIB_PATH=$ALT" "X=1"; do
  env $v timeout -k 10 300 python3 bench.py $B 2> $O/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v%%=*}', round(d['ms_per_step'],2), 'ms', round(d['value'],1))" || { tail -3 $O/bench.err; exit 1; }
done
