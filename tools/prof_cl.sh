#!/bin/bash
# kernel-trace + stats of the bf16 channels-last bench (default surreal-depth1, lanes off so that per-kernel durations add up); summaries in gpurun_out/prof_cl
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
CFG=${1:-surreal-depth1}; PREC=${2:-bf16cl}
rm -rf /tmp/prof_cl; mkdir -p gpurun_out/prof_cl /tmp/prof_cl
DCV_NO_SIDE_STREAMS=1 DCV_CL_NO_WGRAD_SIDE=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_cl -o r --output-format csv -- python3 bench.py --config $CFG --precision $PREC --steps 4 --warmup 2 --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary > gpurun_out/prof_cl/bench_$CFG.log 2>&1
rc=$?
find /tmp/prof_cl -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_cl/kernel_stats_${CFG}_${PREC}.csv \;
grep '^{"metric"' gpurun_out/prof_cl/bench_$CFG.log | cut -c1-200
exit $rc
