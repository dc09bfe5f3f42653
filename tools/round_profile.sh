#!/bin/bash
# Round evidence, everything from ONE box: bench line; rocprofv3 kernel stats + per-dispatch trace of the same command;
# the dominant kernel's PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy - separate passes, no tracing domains mixed in);
# the whole step's per-kernel PMC table; the per-layer tables of the three GPU configs.  Outputs in gpurun_out/round/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export GIT_HEAD=${GIT_HEAD:-$(cat .git_head 2>/dev/null || echo unknown)}
O=gpurun_out/round; rm -rf $O; mkdir -p $O; rm -rf /tmp/rp /tmp/pd_*
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail -3 $O/bench.err; exit 1; }
echo "bench done"; cut -c1-200 $O/bench.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/rp -o r --output-format csv -- python3 bench.py --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary --steps 5 --warmup 2 > $O/bench_under_rocprof.log 2>&1 || { tail -3 $O/bench_under_rocprof.log; exit 1; }
find /tmp/rp -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_lanes.csv \;
find /tmp/rp -name "*kernel_trace.csv" -exec cp {} /tmp/rp/trace.csv \;
# the same command with the discriminators' side streams off: kernels run one at a time, so the per-kernel durations add up to the step
# (with the lanes on, co-running kernels each report the whole overlapped interval)
rm -rf /tmp/rs
DCV_NO_SIDE_STREAMS=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/rs -o r --output-format csv -- python3 bench.py --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary --steps 5 --warmup 2 > $O/bench_under_rocprof_serial.log 2>&1 || { tail -3 $O/bench_under_rocprof_serial.log; exit 1; }
find /tmp/rs -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python3 tools/dominant_from_trace.py /tmp/rp/trace.csv $O/bench_under_rocprof.log > $O/dominant_kernel_trace.json || exit 1
echo "trace done"; cat $O/dominant_kernel_trace.json
pd() { timeout -k 10 300 rocprofv3 --pmc $2 -d /tmp/pd_$1 -o p --output-format csv -- python3 tools/probe_dominant.py 70 10 > $O/pd_$1.log 2>&1 || { tail -3 $O/pd_$1.log; exit 1; }; }
pd FETCH "FETCH_SIZE" && pd WRITE "WRITE_SIZE" && pd SQ "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" || exit 1
python3 tools/dominant_pmc.py > $O/dominant_kernel_pmc.json || exit 1
echo "pmc dominant done"; cat $O/dominant_kernel_pmc.json
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1 || { tail -3 $O/pmc_step.log; exit 1; }
cp gpurun_out/pmc_step/summary.json $O/pmc_step_summary.json; tail -n 10 $O/pmc_step.log
# round 3: one --kernel-trace --stats summary each for the two B = 100 configs and for the 32 x 128 x 128 discriminator stress shape (program directly after --)
for c in surreal-depth1 isogd-flow; do rm -rf /tmp/rc_$c; DCV_NO_SIDE_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/rc_$c -o r --output-format csv -- python3 bench.py --config $c --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary --steps 3 --warmup 1 > $O/bench_under_rocprof_$c.log 2>&1 || { tail -3 $O/bench_under_rocprof_$c.log; exit 1; }
find /tmp/rc_$c -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$c.csv \; ; done
timeout -k 10 200 python3 tools/stress_d.py 4 > $O/stress_d_b4.txt 2>&1 || { tail -3 $O/stress_d_b4.txt; exit 1; }; tail -n 1 $O/stress_d_b4.txt
rm -rf /tmp/rstress; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/rstress -o r --output-format csv -- python3 tools/stress_d.py 4 > $O/stress_d_under_rocprof.log 2>&1 || { tail -3 $O/stress_d_under_rocprof.log; exit 1; }
find /tmp/rstress -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_stress_d_b4.csv \;
for c in isogd-depth surreal-depth1 isogd-flow; do timeout -k 10 250 python3 tools/layer_table.py $c --csv $O/layers_$c.csv > $O/layers_$c.txt 2>&1 || exit 1; tail -n 1 $O/layers_$c.txt; done
for c in surreal-depth1 isogd-flow; do timeout -k 10 300 python3 bench.py --config $c --steps 6 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err || { tail -3 $O/bench_$c.err; exit 1; }; cut -c1-200 $O/bench_$c.json; done
