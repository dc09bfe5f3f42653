#!/bin/bash
# A/B of the library built WITH packed-FP32 VALU instructions (dcvgan_amd/_packed_fp32.so: DCV_PACKED_FP32=1 DCV_OUT=... DCV_OBJ=... bash dcvgan_amd/csrc/build.sh,
# built on the CPU side and carried along) against the shipped build (without them), inside ONE gpurun call.  Counts, per build:
#   harness  tools/race_probe.py   forward + backward of two / three discriminators on their own streams, every gradient compared bit for bit with the one-stream pass
#   step     tools/repeat_probe.py whole iterations (side streams on), 12 seeded runs of 2 iterations: distinct final states
# usage (on the GPU box): bash tools/packed_fp32_ab.sh > gpurun_out/r04_packed_fp32/ab.txt
cd "$GRAFT_REPO_ROOT" || exit 1
SO=dcvgan_amd/libdcvgan_hip.so
cp $SO /tmp/shipped.so
trap 'cp /tmp/shipped.so $SO' EXIT
h() { timeout -k 10 300 python3 tools/race_probe.py surreal-depth1 16 "$@" 2>&1 < /dev/null | grep 'trials' | cut -c1-220; }
s() { timeout -k 10 300 python3 tools/repeat_probe.py surreal-depth1 16 "$@" 2>&1 < /dev/null | grep SUMMARY | cut -c1-160; }
for b in packed shipped; do
  if [ $b = packed ]; then cp dcvgan_amd/_packed_fp32.so $SO; else cp /tmp/shipped.so $SO; fi
  echo "=== build: $b ($( [ $b = packed ] && echo 'with v_pk_*_f32' || echo 'no packed-FP32 instructions'))"
  echo "harness, vdis at bf16 products + gdis fp32, two streams, 150 trials:";   h bf16 150 bwd -q tap only=vdis lanes=vdis,gdis nog
  echo "harness, gdis at bf16 products + idis, vdis fp32, three streams + generators on the main stream, 150 trials:"; h bf16 150 bwd -q tap only=gdis
  echo "harness, everything fp32, three streams + generators, 150 trials:";      h fp32 150 bwd -q tap
  echo "harness, everything at bf16 products, 150 trials:";                      h bf16 150 bwd -q tap
  echo "harness, everything at f32x6, 150 trials:";                              h f32x6 150 bwd -q tap
  [ -n "$HARNESS_ONLY" ] && continue
  for m in fp32 bf16 f32x6; do echo "step, $m, 12 runs x 2 iterations:"; s $m 12 1 2; done
  echo "step, gdis at bf16 products, rest fp32:"; s bf16 12 1 2 only=gdis
done
