#!/bin/bash
# Diagnostic build with cycle stamps in the GEMM kernels (-DDCV_STAMP) for tools/stamps*.py.  Built ON the GPU box into /tmp (it is
# never part of the repo or of the shipped library): gpurun -- 'bash tools/build_stamp.sh && python3 tools/stamps_wg.py'
set -euo pipefail
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/libdcvgan_hip_stamp.so}
O=/tmp/dcv_stamp_obj; mkdir -p $O
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -Idcvgan_amd/csrc -mllvm -amdgpu-mfma-vgpr-form -DDCV_STAMP"
hipcc $F -c dcvgan_amd/csrc/conv_mfma.hip -o $O/conv_mfma.o 2> $O/conv.log &
hipcc $F -c dcvgan_amd/csrc/elementwise.hip -o $O/elementwise.o 2> $O/ew.log &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $O/conv_mfma.o $O/elementwise.o
echo "built $OUT"
