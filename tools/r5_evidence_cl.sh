#!/bin/bash
# round 5 evidence for the 16-bit channels-last path, one box: PMC passes, rocprofv3 kernel stats (lanes off), layer table, bench lines of the three configs, stress shape
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export GIT_HEAD=$(cat .git_head 2>/dev/null || echo unknown)
O=gpurun_out/r5ev_cl; mkdir -p $O
bash tools/pmc_step.sh surreal-depth1 bf16cl $O/pmc > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
tail -n 10 $O/pmc.log
bash tools/prof_cl.sh surreal-depth1 bf16cl > $O/prof_cl.log 2>&1 || { tail -3 $O/prof_cl.log; exit 1; }
cp gpurun_out/prof_cl/kernel_stats_surreal-depth1_bf16cl.csv $O/
timeout -k 10 250 python3 tools/layer_table.py surreal-depth1 --precision bf16cl --csv $O/layers_bf16cl_surreal-depth1.csv > $O/layers.txt 2>&1 || exit 1; tail -n 1 $O/layers.txt
for c in surreal-depth1 isogd-depth isogd-flow; do
  timeout -k 10 200 python3 bench.py --config $c --precision bf16cl --steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer > $O/bench_bf16cl_$c.json 2> $O/bench_$c.err || { tail -3 $O/bench_$c.err; exit 1; }
  cut -c1-150 $O/bench_bf16cl_$c.json
done
for m in bf16cl fp16cl; do for b in 4 100; do timeout -k 10 200 python3 tools/stress_d.py $b $m >> $O/stress_d.txt 2>&1 || { tail -3 $O/stress_d.txt; exit 1; }; done; done
grep -v amdgpu.ids $O/stress_d.txt | tail -n 12
