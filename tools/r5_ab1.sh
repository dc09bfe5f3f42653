#!/bin/bash
# round 5 A/B (one box): wgrad XCD map vs flat ids; gather LDS stages 2 / 3 / 4 — bf16cl layer tables + whole iterations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_cl16_gpu.py -m gpu -x -q > $O/test_cl16_s2.log 2>&1 || { tail -15 $O/test_cl16_s2.log; exit 1; }
tail -n 2 $O/test_cl16_s2.log
DCV_CL_STAGES=3 timeout -k 10 500 python3 -m pytest tests/test_cl16_gpu.py -m gpu -x -q > $O/test_cl16_s3.log 2>&1 || { tail -15 $O/test_cl16_s3.log; exit 1; }
tail -n 2 $O/test_cl16_s3.log
lt() { env $1 timeout -k 10 250 python3 tools/layer_table.py surreal-depth1 --precision bf16cl --csv $O/layers_$2.csv > $O/layers_$2.txt 2>&1 || { tail -5 $O/layers_$2.txt; exit 1; }; echo "$2: $(tail -n 1 $O/layers_$2.txt)"; }
lt DCV_CL_WGRAD_FLAT=1 flat_s2 && lt X=1 xcd_s2 && lt DCV_CL_STAGES=3 xcd_s3 && lt DCV_CL_STAGES=4 xcd_s4 || exit 1
B="--config surreal-depth1 --precision bf16cl --steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary"
for v in "DCV_CL_WGRAD_FLAT=1" "X=1" "DCV_CL_STAGES=3" "DCV_CL_STAGES=4" "DCV_CL_WGRAD_FLAT=1" "X=1" "DCV_CL_STAGES=3"; do
  env $v timeout -k 10 200 python3 bench.py $B 2> $O/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), 'ms', round(d['value'],1))" || { tail -3 $O/bench.err; exit 1; }
done
