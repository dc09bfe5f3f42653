#!/bin/bash
# Build libdcvgan_hip.so for gfx950 (in-tree; the .so travels to the GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../libdcvgan_hip.so"
# -amdgpu-mfma-vgpr-form: MFMA accumulators in architectural VGPRs also in the one-wave-per-SIMD kernels (wgrad_dma_kernel): with the
# default heuristic those get AGPR accumulators, and the compiler moved all 64 of them to VGPRs and back on every tile of the K loop
# (the conditional fold of the two-level accumulation is VALU code): 128 v_accvgpr moves per 128 MFMAs.
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$ROOT/include -I$HERE -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p "$HERE/obj"
pids=()
for f in conv_mfma elementwise conv_cl16 cl_elementwise; do
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$HERE/dcv_common.h" -nt "$HERE/obj/$f.o" ] || [ "$ROOT/include/dcvgan_hip.h" -nt "$HERE/obj/$f.o" ]; then
    hipcc $FLAGS ${EXTRA_HIPCC_FLAGS:-} -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE/obj/conv_mfma.o" "$HERE/obj/elementwise.o" "$HERE/obj/conv_cl16.o" "$HERE/obj/cl_elementwise.o"
echo "built $OUT"
