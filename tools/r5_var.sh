#!/bin/bash
# run-to-run spread of the shipped schedule: N bench runs per variant, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5an; mkdir -p $O
for cfg in isogd-depth surreal-depth1; do
B="--config $cfg --precision bf16cl --steps 12 --warmup 4 --no-cpu-baseline --no-as-trainer --no-minimal --no-secondary"
for r in 1 2 3 4 5 6; do for v in X=1 DCV_CL_WGRAD_HOLD=1 DCV_CL_NO_WGRAD_SIDE=1; do
  env $v timeout -k 10 200 python3 bench.py $B 2> $O/bench.err | V="$v" C="$cfg" python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['C'], os.environ['V'][:24], round(d['ms_per_step'],2), 'ms', round(d.get('peak_mem_gb',0),2), 'GB')" || { tail -3 $O/bench.err; exit 1; }
done; done; done
