#!/bin/bash
# round 5 A/B (one box): gather tiles — DCV_CL_TILES 0 (128 x 128 only), 1 (+ 128 x 256 on 8 waves), 2 (+ 96 x 256), 3 (both)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
for t in 3 0; do DCV_CL_TILES=$t timeout -k 10 500 python3 -m pytest tests/test_cl16_gpu.py -m gpu -x -q > $O/test_cl16_t$t.log 2>&1 || { tail -25 $O/test_cl16_t$t.log; exit 1; }; tail -n 1 $O/test_cl16_t$t.log; done
lt() { env $1 timeout -k 10 250 python3 tools/layer_table.py surreal-depth1 --precision bf16cl --csv $O/layers_$2.csv > $O/layers_$2.txt 2>&1 || { tail -5 $O/layers_$2.txt; exit 1; }; echo "$2: $(tail -n 1 $O/layers_$2.txt)"; }
lt DCV_CL_TILES=0 t0 && lt DCV_CL_TILES=1 t1 && lt DCV_CL_TILES=2 t2 && lt DCV_CL_TILES=3 t3 || exit 1
B="--config surreal-depth1 --precision bf16cl --steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary"
for v in 0 1 2 3 0 1 2 3; do
  DCV_CL_TILES=$v timeout -k 10 200 python3 bench.py $B 2> $O/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('tiles $v', round(d['ms_per_step'],2), 'ms', round(d['value'],1))" || { tail -3 $O/bench.err; exit 1; }
done
