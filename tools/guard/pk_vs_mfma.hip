// Stand-alone attempt at the defect of DESIGN §8(d): a VALU kernel whose arithmetic is packed FP32 (v_pk_fma_f32 / v_pk_mul_f32 with SGPR weight operands, as thin_quad_kernel's
// was) checks itself against the same sums formed with single FP32 instructions, while a second kernel keeps the matrix pipes busy with bf16 (or fp32) MFMAs on another stream.
// All values are small integers, so every order of summation gives the same float: any mismatch is a wrong instruction result, not rounding.
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/guard/pk_vs_mfma.hip -o tools/guard/pk_vs_mfma.so      (built WITH packed FP32: that is the point)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// weights: wave-uniform (scalar loads), data: per lane
__global__ __launch_bounds__(256) void pk_check_kernel(const float* __restrict__ x, const float* __restrict__ w, int n_rows, int iters, unsigned long long* bad, uint32_t* first) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int r = 0; r < n_rows; ++r) {
            const int row = (wave * 7 + r * 3 + it) % n_rows;
            const float a = x[(size_t)row * 64 + lane], b = x[(size_t)((row + 1) % n_rows) * 64 + lane];
            // neighbours as thin_quad_kernel takes them (whole-wave DPP shifts)
            const float ar = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x130, 0xf, 0xf, true));
            const float bl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x138, 0xf, 0xf, true));
            const f32x4* wq = reinterpret_cast<const f32x4*>(w + (size_t)((r + it) % 64) * 16);      // wave-uniform address: scalar loads
            const f32x4 w0 = wq[0], w1 = wq[1], w2 = wq[2], w3 = wq[3];
            // packed form (the compiler pairs these)
            acc0 += (f32x2){a, ar} * (f32x2){w0[0], w0[1]} + (f32x2){b, bl} * (f32x2){w1[0], w1[1]};
            acc1 += (f32x2){ar, a} * (f32x2){w2[2], w2[3]} + (f32x2){bl, b} * (f32x2){w3[2], w3[3]};
            // the same sums with single instructions, kept apart from the vectoriser by an opaque copy
            float a_ = a, ar_ = ar, b_ = b, bl_ = bl;
            asm volatile("" : "+v"(a_), "+v"(ar_), "+v"(b_), "+v"(bl_));
            s0 = __builtin_fmaf(a_, w0[0], s0); s0 = __builtin_fmaf(b_, w1[0], s0);
            s1 = __builtin_fmaf(ar_, w0[1], s1); s1 = __builtin_fmaf(bl_, w1[1], s1);
            s2 = __builtin_fmaf(ar_, w2[2], s2); s2 = __builtin_fmaf(bl_, w3[2], s2);
            s3 = __builtin_fmaf(a_, w2[3], s3); s3 = __builtin_fmaf(b_, w3[3], s3);
        }
        if (acc0[0] != s0 || acc0[1] != s1 || acc1[0] != s2 || acc1[1] != s3) {
            if (nbad == 0 && atomicAdd(bad, 0ull) == 0) { first[0] = blockIdx.x; first[1] = threadIdx.x; first[2] = __builtin_bit_cast(uint32_t, acc0[1]); first[3] = __builtin_bit_cast(uint32_t, s1); }
            ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <bool BF>
__global__ __launch_bounds__(256) void mfma_spin_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float v = (float)((threadIdx.x & 7) - 3);
    bf16x8 a8, b8;
    for (int q = 0; q < 8; ++q) { a8[q] = (__bf16)v; b8[q] = (__bf16)(v + 1.f); }
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_setprio(2);
        for (int i = 0; i < 4; ++i) {
            if constexpr (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, v + 1.f, acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;     // keep the work
}

extern "C" int pk_check_launch(const void* x, const void* w, int n_rows, int iters, void* bad, void* first, int blocks, void* stream) {
    hipLaunchKernelGGL(pk_check_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)w, n_rows, iters, (unsigned long long*)bad, (uint32_t*)first);
    return (int)hipGetLastError();
}
extern "C" int mfma_spin_launch(int bf16, void* out, int iters, int blocks, void* stream) {
    if (bf16) hipLaunchKernelGGL(mfma_spin_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)out, iters);
    else hipLaunchKernelGGL(mfma_spin_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)out, iters);
    return (int)hipGetLastError();
}
