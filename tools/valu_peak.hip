// Sustained rate of v_fma_f32 against v_pk_fma_f32 on one MI355X (is the packed form twice the scalar one?).
// hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_scalar(float* out, float a, float b, int iters) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));   // (plain C is auto-vectorised into v_pk_fma_f32)
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_packed(float* out, float a, float b, int iters) {
    f32x2 x[16];
    const f32x2 av = {a, a * 1.0001f}, bv = {b, b + 1e-3f};
    for (int i = 0; i < 16; ++i) x[i] = f32x2{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i][0] + x[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4 * 256 * 4096);
    const int iters = 20000, blocks = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-3f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("v_fma_f32    : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 16 * iters * 256.0 * blocks / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 1e-3f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("v_pk_fma_f32 : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 32 * iters * 256.0 * blocks / ms / 1e9);
    }
    return 0;
}
