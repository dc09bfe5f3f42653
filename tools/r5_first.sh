#!/bin/bash
# round 5, first call: counters for the bf16 channels-last path + same-box baselines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
bash tools/pmc_step.sh surreal-depth1 bf16cl $O/pmc_bf16cl > $O/pmc_bf16cl.log 2>&1 || { tail -5 $O/pmc_bf16cl.log; exit 1; }
tail -n 12 $O/pmc_bf16cl.log
bash tools/prof_cl.sh surreal-depth1 bf16cl > $O/prof_cl.log 2>&1 || { tail -3 $O/prof_cl.log; exit 1; }
cp gpurun_out/prof_cl/kernel_stats_surreal-depth1_bf16cl.csv $O/
timeout -k 10 250 python3 tools/layer_table.py surreal-depth1 --precision bf16cl --csv $O/layers_bf16cl_surreal-depth1.csv > $O/layers_bf16cl_surreal-depth1.txt 2>&1 || exit 1; tail -n 1 $O/layers_bf16cl_surreal-depth1.txt
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail -3 $O/bench.err; exit 1; }
cut -c1-200 $O/bench.json
