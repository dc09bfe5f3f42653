// Sustained v_mfma_f32_32x32x2_f32 rate on MI355X: pure register loop (no memory), W waves per SIMD,
// then the same loop with 4 ds_read_b32 per 4 MFMAs (the gather kernel's inner step).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float sm[16 * 320];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 320; i += 256) sm[i] = (float)(i % 7) * 0.25f - 0.5f;
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = lane * 0.001f, y = 1.0f - lane * 0.002f, z = 0.5f, w = -0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (LDS) {
                const int kk = (2 * ks + (lane >> 5)) * 320 + (lane & 31);
                x = sm[kk]; y = sm[kk + 32]; z = sm[kk + 64]; w = sm[kk + 96];
            }
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, z, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, w, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, z, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, w, a3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int LDS>
void run(int blocks_per_cu, float* d) {
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<LDS>, dim3(grid), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<LDS>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 32 * (2.0 * 32 * 32 * 2);
    printf("lds=%d waves/SIMD=%d: %.1f ms  %.1f TFLOP/s\n", LDS, blocks_per_cu, ms, flops / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int b = 1; b <= 3; ++b) run<0>(b, d);
    for (int b = 1; b <= 3; ++b) run<1>(b, d);
    return 0;
}
