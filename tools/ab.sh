#!/bin/bash
# ONE A/B driver for a gpurun call (boxes differ by a few per cent, so both arms run on the same box, alternating).
#   tools/ab.sh env "VAR=1 [VAR2=x ...]" [pairs] [bench args...]     the shipped library, environment unset ("base") against set ("alt")
#   tools/ab.sh lib path/to/alt.so        [pairs] [bench args...]     two builds of the library: the shipped .so against an alternative built beforehand
#                                                                      (DCV_OUT=dcvgan_amd/alt_x.so DCV_OBJ=/tmp/alt_obj EXTRA_HIPCC_FLAGS=-D... bash dcvgan_amd/csrc/build.sh)
# AB_TESTS="pytest targets" runs those against the alt arm first (and stops on a failure); AB_LAYERS="filter" adds the per-layer table of both arms;
# AB_OUT names the record under gpurun_out/ (default ab).  Every number is bench.py's own `ms_per_step` (headline schedule only, no secondary legs).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
MODE=$1; WHAT=$2; P=${3:-3}; shift; shift; shift
O=gpurun_out/${AB_OUT:-ab}; mkdir -p "$O"
case "$MODE" in
  env) ALT="$WHAT" ;;
  lib) [ -f "$WHAT" ] || { echo "ab.sh: no library at $WHAT"; exit 1; }; ALT="DCV_LIB_PATH=$(cd "$(dirname "$WHAT")" && pwd)/$(basename "$WHAT")" ;;
  *) echo "usage: ab.sh env|lib <what> [pairs] [bench args]"; exit 1 ;;
esac
BASE="DCV_AB_ARM=base"
bench() { env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary --steps 12 --warmup 3 "${@:2}" 2>"$O/bench.err" \
          | python3 -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
if [ -n "$AB_TESTS" ]; then env $ALT timeout -k 10 1000 python3 -m pytest $AB_TESTS -x -q -m gpu > "$O/tests.log" 2>&1 || { tail -n 30 "$O/tests.log"; exit 1; }; tail -n 1 "$O/tests.log"; fi
if [ -n "${AB_LAYERS+x}" ]; then
  for arm in base alt; do
    [ $arm = base ] && E="$BASE" || E="$ALT"
    env $E timeout -k 10 400 python3 tools/layer_table.py ${AB_CONFIG:-isogd-depth} --filter "$AB_LAYERS" --csv "$O/layers_$arm.csv" > "$O/layers_$arm.txt" 2>&1 || { tail -n 5 "$O/layers_$arm.txt"; exit 1; }
    echo "$arm: $(tail -n 1 "$O/layers_$arm.txt")"
  done
fi
for i in $(seq 1 "$P"); do
  a=$(bench "$BASE" "$@") || { tail -n 5 "$O/bench.err"; exit 1; }
  b=$(bench "$ALT" "$@") || { tail -n 5 "$O/bench.err"; exit 1; }
  echo "pair $i: base $a ms | alt [$WHAT] $b ms" | tee -a "$O/pairs.txt"
done
