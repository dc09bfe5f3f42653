#!/bin/bash
# split-K of the gather kernel for single-class calls with few position tiles (DCV_CL_NO_SPLITK=1 = off)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5r; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_cl16_gpu.py tests/test_cl16_b100_gpu.py tests/test_fp16_gpu.py tests/test_cl16_oracle_gpu.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -25 $O/tests.log; exit 1; }
tail -n 2 $O/tests.log
lt() { env $1 timeout -k 10 250 python3 tools/layer_table.py surreal-depth1 --precision bf16cl --csv $O/layers_$2.csv > $O/layers_$2.txt 2>&1 || { tail -5 $O/layers_$2.txt; exit 1; }; echo "$2: $(tail -n 1 $O/layers_$2.txt)"; }
lt DCV_CL_NO_SPLITK=1 off && lt X=1 on || exit 1
B="--config surreal-depth1 --precision bf16cl --steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary"
for v in "DCV_CL_NO_SPLITK=1" "X=1" "DCV_CL_NO_SPLITK=1" "X=1" "DCV_CL_NO_SPLITK=1" "X=1"; do
  env $v timeout -k 10 200 python3 bench.py $B 2> $O/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), 'ms', round(d['value'],1))" || { tail -3 $O/bench.err; exit 1; }
done
