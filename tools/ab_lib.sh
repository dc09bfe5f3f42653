#!/bin/bash
# A/B of two BUILDS of the library inside ONE gpurun call: the shipped .so ("base") against a build with extra hipcc flags ("alt",
# e.g. -DDCV_PRO_PRIO=3), swapped in place between alternating bench runs.   usage: ab_lib.sh "<extra hipcc flags>" [pairs] [bench args...]
# With DCV_AB_TESTS set, that pytest selection runs against the alt build first.  The alt build goes through dcvgan_amd/csrc/build.sh (same flags, all
# four translation units) into /tmp; DCV_PACKED_FP32=1 in the environment builds it WITH packed-FP32 instructions (tools/packed_fp32_ab.sh is the full A/B of that).
cd "$GRAFT_REPO_ROOT" || exit 1
X=$1; P=${2:-3}; shift; shift
SO=dcvgan_amd/libdcvgan_hip.so
cp $SO /tmp/base.so
trap 'cp /tmp/base.so $SO' EXIT      # whatever ends the script (a failed bench, a timeout, an interrupt), the shipped build is back in place
EXTRA_HIPCC_FLAGS="$X" DCV_OUT=/tmp/alt.so DCV_OBJ=/tmp/ab_obj bash dcvgan_amd/csrc/build.sh > /tmp/ab_build.log 2>&1 || { tail /tmp/ab_build.log; exit 1; }
bench() { timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-minimal --no-as-trainer --no-secondary --steps 12 --warmup 3 "$@" 2>/dev/null | python3 -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])"; }
if [ -n "$DCV_AB_TESTS" ]; then cp /tmp/alt.so $SO; timeout -k 10 900 python3 -m pytest $DCV_AB_TESTS -x -q -m gpu 2>&1 | tail -n 3; fi
for i in $(seq 1 $P); do
cp /tmp/base.so $SO; a=$(bench "$@") || exit 1
cp /tmp/alt.so $SO; b=$(bench "$@") || exit 1
echo "pair $i: base $a ms | alt [$X] $b ms"
done
cp /tmp/base.so $SO
