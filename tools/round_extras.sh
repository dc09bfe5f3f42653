#!/bin/bash
# Second evidence call of a round: the whole GPU suite (gradient-parity reports into gpurun_out/round/parity), smoke(), the
# bf16 throughput mode's bench lines and layer table.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/round; mkdir -p $O/parity
DCV_REPORT_DIR=$O/parity timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -n 30 $O/pytest_gpu.log; exit 1; }
tail -n 2 $O/pytest_gpu.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -n 20 $O/smoke.log; exit 1; }
tail -n 2 $O/smoke.log
for c in isogd-depth surreal-depth1 isogd-flow; do timeout -k 10 300 python3 bench.py --config $c --precision bf16 --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_bf16_$c.json 2> $O/bench_bf16_$c.err || { tail -3 $O/bench_bf16_$c.err; exit 1; }; cut -c1-200 $O/bench_bf16_$c.json; done
timeout -k 10 250 python3 tools/layer_table.py isogd-depth --precision bf16 --csv $O/layers_isogd-depth_bf16.csv > $O/layers_isogd-depth_bf16.txt 2>&1 || exit 1; tail -n 1 $O/layers_isogd-depth_bf16.txt
