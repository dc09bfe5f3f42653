#!/bin/bash
# A/B of one environment toggle inside ONE gpurun call (boxes differ by a few per cent): usage  ab_env.sh DCV_NO_XYZ [pytest targets...]
# ops/b70 parity tests first, then the per-layer table with the toggle off / on, then two alternating bench pairs.
cd "$GRAFT_REPO_ROOT" || exit 1
T=$1; shift
O=gpurun_out/ab_$T; mkdir -p $O
timeout -k 10 600 python3 -m pytest ${@:-tests/test_ops_gpu.py tests/test_b70_gpu.py} -x -q -m gpu > $O/tests.log 2>&1 || { tail -n 30 $O/tests.log; exit 1; }
tail -n 2 $O/tests.log
timeout -k 10 200 python3 tools/layer_table.py isogd-depth --csv $O/layers_new.csv > $O/layers_new.txt 2>&1 && tail -n 1 $O/layers_new.txt &&
env $T=1 timeout -k 10 200 python3 tools/layer_table.py isogd-depth --csv $O/layers_old.csv > $O/layers_old.txt 2>&1 && tail -n 1 $O/layers_old.txt &&
for i in 1 2; do
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | cut -c1-120 &&
env $T=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | cut -c1-120 || exit 1; done
