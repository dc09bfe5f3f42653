#!/bin/bash
# kernel-trace + stats of the bench command; summaries land in gpurun_out/${1:-prof_stats}   (extra bench args after the name)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
N=${1:-prof_stats}; shift
rm -rf /tmp/prof_stats; mkdir -p gpurun_out/$N /tmp/prof_stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r --output-format csv -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary "$@" > gpurun_out/$N/bench.log 2>&1
rc=$?
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/$N/ \;
python3 tools/gaps.py /tmp/prof_stats > gpurun_out/$N/gaps.txt 2>&1; python3 tools/topk.py /tmp/prof_stats > gpurun_out/$N/topk.txt 2>&1
python3 tools/timeline.py /tmp/prof_stats > gpurun_out/$N/timeline.txt 2>&1
grep '^{"metric"' gpurun_out/$N/bench.log | cut -c1-200
exit $rc
