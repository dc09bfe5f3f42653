#!/bin/bash
# Round-4 evidence beside tools/round_profile.sh (one box, one call): the bf16 channels-last path (bench lines of the three GPU configs, rocprofv3 kernel stats
# and the layer table of surreal-depth1, the stress shape, the tolerance reports of tests/test_cl16_gpu.py), the fp32-on-bf16 mode (probe, layer table, bench
# line) and the as-trainer legs of the default bench.  Outputs in gpurun_out/round4/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/round4; mkdir -p $O
B="--no-cpu-baseline --no-as-trainer"
for c in surreal-depth1 isogd-depth isogd-flow; do
  timeout -k 10 200 python3 bench.py --config $c --precision bf16cl --steps 12 --warmup 4 $B > $O/bench_bf16cl_$c.json 2> $O/bench_bf16cl_$c.err || { tail -3 $O/bench_bf16cl_$c.err; exit 1; }
  cut -c1-160 $O/bench_bf16cl_$c.json
done
bash tools/prof_cl.sh surreal-depth1 bf16cl > $O/prof_cl.log 2>&1 || { tail -3 $O/prof_cl.log; exit 1; }
cp gpurun_out/prof_cl/kernel_stats_surreal-depth1_bf16cl.csv $O/kernel_stats_bf16cl_surreal-depth1.csv
timeout -k 10 250 python3 tools/layer_table.py surreal-depth1 --precision bf16cl --csv $O/layers_bf16cl_surreal-depth1.csv > $O/layers_bf16cl_surreal-depth1.txt 2>&1 || exit 1; tail -n 1 $O/layers_bf16cl_surreal-depth1.txt
for b in 4 100; do timeout -k 10 200 python3 tools/stress_d.py $b bf16cl >> $O/stress_d_bf16cl.txt 2>&1 || exit 1; done
timeout -k 10 200 python3 tools/stress_d.py 4 fp32 >> $O/stress_d_bf16cl.txt 2>&1; tail -n 3 $O/stress_d_bf16cl.txt
timeout -k 10 400 python3 -m pytest tests/test_cl16_gpu.py -m gpu -q > $O/test_cl16.log 2>&1 || { tail -5 $O/test_cl16.log; exit 1; }
cat gpurun_out/cl16_iteration_*.txt > $O/cl16_tolerance.txt; for f in gpurun_out/cl16_models_*.txt; do echo "== $f (quarter-width models, CL16 path vs fp32 path; control = the bf16-product mode)" >> $O/cl16_tolerance.txt; cat $f >> $O/cl16_tolerance.txt; done
tail -n 2 $O/test_cl16.log
timeout -k 10 300 python3 tools/f32x6_probe.py 70 > $O/f32x6_probe.txt 2>&1 || { tail -3 $O/f32x6_probe.txt; exit 1; }; grep speed-up $O/f32x6_probe.txt
timeout -k 10 250 python3 tools/layer_table.py isogd-depth --precision f32x6 --csv $O/layers_f32x6_isogd-depth.csv > $O/layers_f32x6.txt 2>&1 || exit 1; tail -n 1 $O/layers_f32x6.txt
timeout -k 10 200 python3 bench.py --precision f32x6 --steps 15 --warmup 4 $B > $O/bench_f32x6.json 2> $O/bench_f32x6.err || { tail -3 $O/bench_f32x6.err; exit 1; }; cut -c1-160 $O/bench_f32x6.json
