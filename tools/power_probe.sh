#!/bin/bash
# Is the step power-limited?  Sample rocm-smi (socket power, sclk, temperature) every 0.3 s while the bench runs 80 iterations.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/power; mkdir -p $O
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-minimal --steps 80 --warmup 3 "$@" > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 6
for i in $(seq 1 25); do rocm-smi --showpower --showclocks --showtemp --showmaxpower 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)|Max Graphics" | tr '\n' ' ' ; echo; sleep 0.3; kill -0 $BP 2>/dev/null || break; done > $O/samples.txt
wait $BP
cut -c1-160 $O/bench.json; head -3 $O/samples.txt; echo ...; tail -n 3 $O/samples.txt
