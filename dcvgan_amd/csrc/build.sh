#!/bin/bash
# Build libdcvgan_hip.so for gfx950 (in-tree; the .so travels to the GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="${DCV_OUT:-$HERE/../libdcvgan_hip.so}"
OBJ="${DCV_OBJ:-$HERE/obj}"
# -amdgpu-mfma-vgpr-form: MFMA accumulators in architectural VGPRs also in the one-wave-per-SIMD kernels (wgrad_dma_kernel): with the
# default heuristic those get AGPR accumulators, and the compiler moved all 64 of them to VGPRs and back on every tile of the K loop
# (the conditional fold of the two-level accumulation is VALU code): 128 v_accvgpr moves per 128 MFMAs.
# -target-feature -packed-fp32-ops: no packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any kernel.  Measured on MI355X
# (profiles/r04_packed_fp32/): waves of the small kernels (thin data gradients, BatchNorm, elementwise) that use them gave wrong upper lanes now
# and then while another stream's bf16-MFMA waves shared their CUs; single-instruction FP32 code of the same kernels did not, in any of the runs.
# (The host pass prints "not a recognized feature for this target" for it; filtered below.)
# DCV_PACKED_FP32=1 builds WITH them (into $DCV_OUT), for the A/B of tools/packed_fp32_ab.sh only.
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
[ -n "${DCV_PACKED_FP32:-}" ] && NOPK=""
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$ROOT/include -I$HERE -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $NOPK"
mkdir -p "$OBJ"
pids=()
for f in conv_mfma elementwise conv_cl16 cl_elementwise; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$HERE/$f.hip" -nt "$OBJ/$f.o" ] || [ "$HERE/dcv_common.h" -nt "$OBJ/$f.o" ] || [ "$ROOT/include/dcvgan_hip.h" -nt "$OBJ/$f.o" ]; then
    hipcc $FLAGS ${EXTRA_HIPCC_FLAGS:-} -c "$HERE/$f.hip" -o "$OBJ/$f.o" 2> >(grep -v "not a recognized feature for this target" >&2) &
    pids+=($!)
  fi
done
# the 16-bit channels-last path a second time with fp16 as its element type (-DDCV_CL_FP16: entry points dcv_clf16_*; BASELINE configs[4] names fp16 MFMA)
for f in conv_cl16 cl_elementwise; do
  if [ ! -f "$OBJ/${f}_f16.o" ] || [ "$HERE/$f.hip" -nt "$OBJ/${f}_f16.o" ] || [ "$HERE/dcv_common.h" -nt "$OBJ/${f}_f16.o" ] || [ "$ROOT/include/dcvgan_hip.h" -nt "$OBJ/${f}_f16.o" ]; then
    hipcc $FLAGS -DDCV_CL_FP16 ${EXTRA_HIPCC_FLAGS:-} -c "$HERE/$f.hip" -o "$OBJ/${f}_f16.o" 2> >(grep -v "not a recognized feature for this target" >&2) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ/conv_mfma.o" "$OBJ/elementwise.o" "$OBJ/conv_cl16.o" "$OBJ/cl_elementwise.o" "$OBJ/conv_cl16_f16.o" "$OBJ/cl_elementwise_f16.o"
# The flag above is only worth something if it took effect (its "not a recognized feature" host-pass message is filtered, and a hipcc that dropped or renamed
# the feature would bring the packed instructions back silently): disassemble the device code of the library just linked and fail on any packed-FP32 arithmetic.
OBJDUMP=/opt/rocm/lib/llvm/bin/llvm-objdump
if [ -z "${DCV_PACKED_FP32:-}" ] && [ -x "$OBJDUMP" ]; then
  TMPD="$(mktemp -d)"; cp "$OUT" "$TMPD/lib.so"
  ( cd "$TMPD" && "$OBJDUMP" --offloading lib.so > /dev/null 2>&1 )
  n=0; found=0
  for co in "$TMPD"/lib.so.*gfx950*; do
    [ -f "$co" ] || continue
    n=$((n + 1))
    if "$OBJDUMP" -d --mcpu=gfx950 "$co" | grep -E -m 5 "v_pk_(fma|mul|add)_f32"; then found=1; fi
  done
  rm -rf "$TMPD"
  [ "$n" -gt 0 ] || { echo "build.sh: could not extract the gfx950 code object from $OUT to check it" >&2; exit 1; }
  [ "$found" -eq 0 ] || { echo "build.sh: packed-FP32 arithmetic (v_pk_*_f32) in the device code although DCV_PACKED_FP32 is unset: the -packed-fp32-ops target feature did not take effect" >&2; rm -f "$OUT"; exit 1; }
fi
echo "built $OUT"
