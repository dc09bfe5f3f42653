#!/bin/bash
# The fp32 parity suites with f32x6 as the PROCESS default (DCV_PRECISION): every LDS-DMA gather kernel then runs the fp32-on-bf16-pipe emulation while the tests keep
# their fp32 bars.  194 of 194 since the sign-alternating accumulation phases; 190 before them.   usage (GPU box): bash tools/x6_parity.sh [log]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
LOG=${1:-gpurun_out/x6_parity.log}; mkdir -p "$(dirname "$LOG")"
DCV_PRECISION=f32x6 timeout -k 10 1100 python3 -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_fullwidth_gpu.py tests/test_b70_gpu.py tests/test_b100_gpu.py \
    tests/test_reference_style_gpu.py tests/test_sampling_gpu.py -m gpu -q > "$LOG" 2>&1
rc=$?
tail -n 3 "$LOG"
exit $rc
