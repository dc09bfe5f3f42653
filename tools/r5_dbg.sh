#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5dbg; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_cl16_gpu.py tests/test_fp16_gpu.py -m gpu -x -q 2>&1 | tail -n 3
rm -rf /tmp/ps; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ps -o r --output-format csv -- python3 tools/stress_d.py 4 bf16cl > $O/stress4.log 2>&1; tail -n 2 $O/stress4.log
find /tmp/ps -name "*kernel_stats.csv" -exec cp {} $O/stress4_stats.csv \;
head -n 14 $O/stress4_stats.csv | cut -c1-150
for v in "DCV_CL_NO_STEM3=1" "X=1" "DCV_CL_NO_STEM3=1" "X=1"; do
  env $v timeout -k 10 200 python3 bench.py --config surreal-depth1 --precision bf16cl --steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary 2> $O/bench.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16cl $v', round(d['ms_per_step'],2), 'ms', round(d['value'],1))" || { tail -3 $O/bench.err; exit 1; }
done
timeout -k 10 250 python3 tools/layer_table.py surreal-depth1 --precision bf16cl --filter "dis." > $O/layers_dis.txt 2>&1; grep "dis\.[gc1] \|gdis.1" $O/layers_dis.txt
