cd "$GRAFT_REPO_ROOT"; O=gpurun_out/quad; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_ops_gpu.py tests/test_b70_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -n 30 $O/tests.log; exit 1; }
tail -n 2 $O/tests.log
timeout -k 10 200 python3 tools/layer_table.py isogd-depth --csv $O/layers_on.csv > $O/layers_on.txt 2>&1 && tail -n 1 $O/layers_on.txt &&
DCV_NO_QUAD=1 timeout -k 10 200 python3 tools/layer_table.py isogd-depth --csv $O/layers_off.csv > $O/layers_off.txt 2>&1 && tail -n 1 $O/layers_off.txt &&
for i in 1 2; do
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | cut -c1-120 &&
DCV_NO_QUAD=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | cut -c1-120 || exit 1; done
