#!/bin/bash
# Round-end evidence: bench line, rocprofv3 kernel stats of the same command, PMC summary.  Outputs in gpurun_out/round/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/round; rm -rf /tmp/rp
timeout -k 10 600 python3 bench.py > gpurun_out/round/bench.json 2> gpurun_out/round/bench.err || { tail -3 gpurun_out/round/bench.err; exit 1; }
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/rp -o r --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/round/bench_under_rocprof.log 2>&1 || { tail -3 gpurun_out/round/bench_under_rocprof.log; exit 1; }
find /tmp/rp -name "*kernel_stats.csv" -exec cp {} gpurun_out/round/kernel_stats.csv \;
bash tools/pmc_step.sh > gpurun_out/round/pmc.log 2>&1 || { tail -3 gpurun_out/round/pmc.log; exit 1; }
cp gpurun_out/pmc_step/summary.json gpurun_out/round/pmc_summary.json
tail -2 gpurun_out/round/pmc.log; cut -c1-300 gpurun_out/round/bench.json
