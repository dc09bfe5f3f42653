// Internal helpers shared by the gfx950 kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "dcvgan_hip.h"

namespace dcv {

extern thread_local char g_err[512];
extern std::atomic<uint64_t> g_launches;

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define DCV_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return dcv::fail(DCV_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define DCV_LAUNCH_CHECK()                                                               \
    do {                                                                                 \
        dcv::g_launches.fetch_add(1, std::memory_order_relaxed);                         \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess)                                                            \
            return dcv::fail(DCV_EHIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline int64_t numel(const dcv_dims5& d) { return (int64_t)d.n * d.c * d.d * d.h * d.w; }
inline bool same_shape(const dcv_dims5& a, const dcv_dims5& b) {
    return a.n == b.n && a.c == b.c && a.d == b.d && a.h == b.h && a.w == b.w;
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// 32-bit magic division (exact for all 0 <= n < 2^31 with 1 <= d < 2^31):
// q = (uint64(n) * mul) >> 32 >> shift
struct FastDiv {
    uint32_t mul, shift, div, pad;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.div = d;
    f.pad = 0;
    if (d == 1) { f.mul = 0; f.shift = 0; return f; }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;   // ceil(log2 d)
    uint64_t m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    f.mul = (uint32_t)m;
    f.shift = l;
    return f;
}

}  // namespace dcv

#ifdef __HIPCC__
// Element type of the 16-bit channels-last path (conv_cl16.hip, cl_elementwise.hip): bf16 by default; -DDCV_CL_FP16 compiles the same two translation units a second
// time for fp16 — same MFMA rate and fragment path (v_mfma_f32_32x32x16_f16), 10 mantissa bits instead of 7, exponent range 6.5e4 instead of fp32's — with the entry
// points renamed dcv_cl_* -> dcv_clf16_* (BASELINE configs[4] names fp16 MFMA; built for the discriminators' stress shape).
namespace dcv {
#ifdef DCV_CL_FP16
typedef _Float16 cl_h;
#define CL_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define CL_HALF_NAME "fp16"
#else
typedef __bf16 cl_h;
#define CL_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define CL_HALF_NAME "bf16"
#endif
typedef cl_h cl_h8 __attribute__((ext_vector_type(8)));
typedef cl_h cl_h2 __attribute__((ext_vector_type(2)));
// two floats -> one dword of two 16-bit values (round to nearest even), and back
__device__ __forceinline__ uint32_t cl_pack2(float a, float b) {
    typedef float f32x2c __attribute__((ext_vector_type(2)));
    const f32x2c t = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(t, cl_h2));
}
#ifdef DCV_CL_FP16
__device__ __forceinline__ float cl_lo(uint32_t w) { return (float)__builtin_bit_cast(cl_h2, w)[0]; }
__device__ __forceinline__ float cl_hi(uint32_t w) { return (float)__builtin_bit_cast(cl_h2, w)[1]; }
#else
__device__ __forceinline__ float cl_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float cl_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
#endif
__device__ __forceinline__ float cl_round(float v) { return (float)(cl_h)v; }      // the value as it will be stored
}  // namespace dcv
namespace dcv {
// n / d for the FastDiv above (d == 1 handled by mul == 0 convention)
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv f) {
    if (f.div == 1) return n;
    uint32_t t = __umulhi(n, f.mul);
    return (t + ((n - t) >> 1)) >> (f.shift - 1);
}
}  // namespace dcv
#endif
