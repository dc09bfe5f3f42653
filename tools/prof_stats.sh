#!/bin/bash
# kernel-trace + stats of the bench command; summaries land in gpurun_out/prof_stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf /tmp/prof_stats; mkdir -p gpurun_out/prof_stats /tmp/prof_stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r --output-format csv -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary > gpurun_out/prof_stats/bench.log 2>&1
rc=$?
find /tmp/prof_stats -name "*stats.csv" -exec cp {} gpurun_out/prof_stats/ \;
python3 tools/gaps.py /tmp/prof_stats > gpurun_out/prof_stats/gaps.txt 2>&1; python3 tools/topk.py /tmp/prof_stats > gpurun_out/prof_stats/topk.txt 2>&1
grep '^{"metric"' gpurun_out/prof_stats/bench.log | cut -c1-200
exit $rc
