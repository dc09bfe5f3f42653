#!/bin/bash
# Round evidence of the 16-bit channels-last path + the default-run bench line + the iteration timeline, one box: outputs in gpurun_out/round_cl/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export GIT_HEAD=${GIT_HEAD:-$(cat .git_head 2>/dev/null || echo unknown)}
O=gpurun_out/round_cl; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 bench.py > $O/bench_default_run.json 2> $O/bench_default_run.err || { tail -3 $O/bench_default_run.err; exit 1; }
cut -c1-200 $O/bench_default_run.json
bash tools/prof_stats.sh round_cl/timeline_run > /dev/null 2>&1; cp gpurun_out/round_cl/timeline_run/timeline.txt $O/timeline.txt 2>/dev/null; head -3 $O/timeline.txt
for c in surreal-depth1 isogd-depth isogd-flow; do timeout -k 10 300 python3 bench.py --config $c --precision bf16cl --steps 8 --warmup 3 --no-cpu-baseline --no-minimal --no-as-trainer > $O/bench_bf16cl_$c.json 2> $O/bench_bf16cl_$c.err || { tail -3 $O/bench_bf16cl_$c.err; exit 1; }; cut -c1-160 $O/bench_bf16cl_$c.json; done
bash tools/pmc_step.sh surreal-depth1 bf16cl gpurun_out/round_cl/pmc_step_bf16cl > $O/pmc_step_bf16cl.log 2>&1 || { tail -3 $O/pmc_step_bf16cl.log; exit 1; }
rm -rf /tmp/rcl; DCV_NO_SIDE_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/rcl -o r --output-format csv -- python3 bench.py --config surreal-depth1 --precision bf16cl --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary --steps 3 --warmup 1 > $O/bench_under_rocprof_bf16cl.log 2>&1 || { tail -3 $O/bench_under_rocprof_bf16cl.log; exit 1; }
find /tmp/rcl -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bf16cl_surreal-depth1.csv \;
timeout -k 10 300 python3 tools/layer_table.py surreal-depth1 --precision bf16cl --csv $O/layers_bf16cl_surreal-depth1.csv > $O/layers_bf16cl.txt 2>&1 || { tail -3 $O/layers_bf16cl.txt; exit 1; }; tail -n 1 $O/layers_bf16cl.txt
