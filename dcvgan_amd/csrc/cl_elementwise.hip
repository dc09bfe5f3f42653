// Elementwise / reduction kernels of the bf16 channels-last ("CL16") data path (see conv_cl16.hip): layout conversion at the module boundary,
// BatchNorm (+ Dropout2d mask) (+ activation) forward and backward, activation derivatives, noise, temporal difference.
// Tensors are "sample-linear": element (n, pixel, c) at n * sn + pixel * pitch + c, pixel = (d, h, w) linear, c contiguous; a thread owns one
// 16-byte group of 8 channels of a pixel, so a wave reads and writes whole 128-byte lines when C >= 64.  Statistics and parameters are fp32.
#include "dcv_common.h"

#ifdef DCV_CL_FP16      // the fp16 build of this file: same code, element type cl_h = _Float16, entry points dcv_clf16_*
#define dcv_cl_from_f32 dcv_clf16_from_f32
#define dcv_cl_to_f32 dcv_clf16_to_f32
#define dcv_cl_elementwise dcv_clf16_elementwise
#define dcv_cl_bn_workspace_bytes dcv_clf16_bn_workspace_bytes
#define dcv_cl_bn_act_forward dcv_clf16_bn_act_forward
#define dcv_cl_bn_act_forward_stats dcv_clf16_bn_act_forward_stats
#define dcv_cl_bn_act_backward dcv_clf16_bn_act_backward
#define cl_bn_finalize_stat_kernel cl_bn_finalize_stat_kernel_f16      // (a kernel defined at file scope inside the extern "C" block)
#endif

#include <algorithm>

namespace dcv {
#ifdef DCV_CL_FP16
inline namespace clf16 {
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct ClView {          // one sample-linear bf16 tensor
    int64_t sn;          // sample stride, elements
    int32_t pitch;       // pixel pitch, elements
};
struct ClShape {
    int32_t N, C, PIX, G;      // samples, channels, pixels per sample, 8-channel groups (ceil(C / 8))
    FastDiv div_g, div_pix;
};

__device__ __forceinline__ void cl_unpack(const u32x4 w, float (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = cl_lo(w[q]); v[2 * q + 1] = cl_hi(w[q]); }
}
__device__ __forceinline__ u32x4 cl_pack(const float (&v)[8]) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    u32x4 w;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x2_ f = {v[2 * q], v[2 * q + 1]}; w[q] = cl_pack2(f[0], f[1]); }
    return w;
}
__device__ __forceinline__ u32x4 cl_ld(const cl_h* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void cl_st(cl_h* p, const u32x4 w) { *reinterpret_cast<u32x4*>(p) = w; }

// work item i -> (n, pixel, group): group fastest
__device__ __forceinline__ void cl_decode(const ClShape& s, int64_t i, int& n, int& pix, int& g) {
    const uint32_t pg = (uint32_t)(i / s.G);      // G is small: 64-bit division only on the host-sized index
    g = (int)(i - (int64_t)pg * s.G);
    n = (int)fdiv(pg, s.div_pix);
    pix = (int)(pg - (uint32_t)n * s.div_pix.div);
}

// ---- fp32 NCDHW (any strides) <-> CL16 ------------------------------------------------------
struct ClCvtArgs {
    const float* f; cl_h* b;
    int64_t f_sn, f_sc; int32_t f_sd, f_sh, f_sw;
    int32_t D, H, W, C, Cpad;       // Cpad: channels written on the CL16 side (zeros past C)
    int64_t b_sn; int32_t b_pitch;
    int64_t total;                  // N * PIX * (Cpad / 8)
    FastDiv div_g, div_pix, div_hw, div_w;
};
__global__ __launch_bounds__(256) void cl_from_f32_kernel(const ClCvtArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    // pixel fastest across lanes (coalesced fp32 plane reads); group slowest within a sample
    const uint32_t G = a.div_g.div, PIX = a.div_pix.div;
    const int64_t per = (int64_t)G * PIX;
    const int n = (int)(i / per);
    const uint32_t r = (uint32_t)(i - (int64_t)n * per);
    const uint32_t g = fdiv(r, a.div_pix), pix = r - g * PIX;
    const uint32_t d = fdiv(pix, a.div_hw);
    uint32_t q = pix - d * a.div_hw.div;
    const uint32_t h = fdiv(q, a.div_w), w = q - h * a.div_w.div;
    const float* src = a.f + (int64_t)n * a.f_sn + (int64_t)d * a.f_sd + (int64_t)h * a.f_sh + (int64_t)w * a.f_sw;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int c = (int)g * 8 + e; v[e] = c < a.C ? src[(int64_t)c * a.f_sc] : 0.f; }
    cl_st(a.b + (int64_t)n * a.b_sn + (int64_t)pix * a.b_pitch + g * 8, cl_pack(v));
}
// accumulate = 1: f += value (gradient accumulation into an fp32 leaf)
__global__ __launch_bounds__(256) void cl_to_f32_kernel(const ClCvtArgs a, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const uint32_t G = a.div_g.div, PIX = a.div_pix.div;
    const int64_t per = (int64_t)G * PIX;
    const int n = (int)(i / per);
    const uint32_t r = (uint32_t)(i - (int64_t)n * per);
    const uint32_t g = fdiv(r, a.div_pix), pix = r - g * PIX;
    const uint32_t d = fdiv(pix, a.div_hw);
    uint32_t q = pix - d * a.div_hw.div;
    const uint32_t h = fdiv(q, a.div_w), w = q - h * a.div_w.div;
    float* dst = const_cast<float*>(a.f) + (int64_t)n * a.f_sn + (int64_t)d * a.f_sd + (int64_t)h * a.f_sh + (int64_t)w * a.f_sw;
    float v[8];
    cl_unpack(cl_ld(a.b + (int64_t)n * a.b_sn + (int64_t)pix * a.b_pitch + g * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = (int)g * 8 + e;
        if (c < a.C) dst[(int64_t)c * a.f_sc] = accumulate ? dst[(int64_t)c * a.f_sc] + v[e] : v[e];
    }
}

// ---- generic elementwise over one or two CL16 tensors ----------------------------------------------
// kind 0: y = x (copy)            1: y = a x + b z           2: y = x + sigma N(0,1)        3: dx = dy * lrelu'(y; slope)   4: dx = dy * (1 - y^2)
// 5: y = lrelu(x)                 6: y = tanh(x)
struct ClEwArgs {
    const cl_h* x; const cl_h* z; cl_h* y;
    ClView xv, zv, yv;
    ClShape s;
    int64_t total;
    int32_t kind; float a, b;
    uint64_t seed, offset;
};
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void cl_normal4(uint64_t seed, uint64_t offset, uint64_t idx, float (&o)[4]) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float s = 2.3283064365386963e-10f;
    const float u0 = ((float)c[0] + 0.5f) * s, u1 = ((float)c[1] + 0.5f) * s, u2 = ((float)c[2] + 0.5f) * s, u3 = ((float)c[3] + 0.5f) * s;
    const float r0 = sqrtf(-2.f * __logf(fmaxf(u0, 1e-30f))), r1 = sqrtf(-2.f * __logf(fmaxf(u2, 1e-30f)));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    o[0] = r0 * c0; o[1] = r0 * s0; o[2] = r1 * c1; o[3] = r1 * s1;
}
__global__ __launch_bounds__(256) void cl_ew_kernel(const ClEwArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    int n, pix, g;
    cl_decode(a.s, i, n, pix, g);
    float x[8], z[8], y[8];
    cl_unpack(cl_ld(a.x + (int64_t)n * a.xv.sn + (int64_t)pix * a.xv.pitch + g * 8), x);
    if (a.kind == 1 || a.kind == 3 || a.kind == 4) cl_unpack(cl_ld(a.z + (int64_t)n * a.zv.sn + (int64_t)pix * a.zv.pitch + g * 8), z);
    if (a.kind == 2) {
        float t[4];
        cl_normal4(a.seed, a.offset, (uint64_t)i * 2, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = t[e];
        cl_normal4(a.seed, a.offset, (uint64_t)i * 2 + 1, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) z[4 + e] = t[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool real = g * 8 + e < a.s.C;        // padding channels stay zero
        float v;
        switch (a.kind) {
            case 0: v = x[e]; break;
            case 1: v = a.a * x[e] + a.b * z[e]; break;
            case 2: v = x[e] + a.a * z[e]; break;
            case 3: v = x[e] * (z[e] > 0.f ? 1.f : a.a); break;
            case 4: v = x[e] * (1.f - z[e] * z[e]); break;
            case 5: v = x[e] > 0.f ? x[e] : x[e] * a.a; break;
            default: v = tanhf(x[e]); break;
        }
        y[e] = real ? v : 0.f;
    }
    cl_st(a.y + (int64_t)n * a.yv.sn + (int64_t)pix * a.yv.pitch + g * 8, cl_pack(y));
}

// ---- BatchNorm ------------------------------------------------------------------------------
// Block = PB pixels x G groups (G = C / 8 <= 128): thread (p, g) walks pixels p, p + PB * gridDim.x, ...; per-thread fp32 sums of its 8 channels
// (bounded runs: at most `run` terms, then folded into doubles), block-combined through LDS in a fixed order, written as partial[block][C][NV].
struct ClBnArgs {
    const cl_h* x; const cl_h* dy; cl_h* y;
    ClView xv, dyv, yv;
    int32_t N, C, PIX, G;
    FastDiv div_pix;
    int64_t P;                       // N * PIX
    const float* scale; const float* shift;      // y = act(mask * (x * scale + shift))
    const float* mask;               // N * C or null
    const float* c1; const float* c2; const float* c3;   // backward: dx = c1 * dz - c2 * xhat' ... (see cl_bn_bwd_apply)
    const float* gamma; const float* beta; const float* mean; const float* invstd;   // backward kernels form scale / shift themselves (one launch less)
    double* partial;                 // [blocks][C][NV]
    int32_t act; float slope;
    int32_t pb, pad;                 // pixels per block iteration
};
template <int NV>
__device__ __forceinline__ void cl_bn_block_out(double (&s)[8][NV], const ClBnArgs& a, int g, int p, int PB, double* red) {
    // combine the block's PB pixel slots per channel, in a fixed order, and write partial[block][C][NV]
    if ((a.G & (a.G - 1)) == 0 && a.G <= 64) {
        // G | 64: thread p * G + g sits on lane (p * G + g) % 64, so the slots of one group within a wave are G lanes apart: xor tree over the lane bits
        // above log2(G), then the four waves meet in LDS (red: [4][G * 8][NV])
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int v = 0; v < NV; ++v)
                for (int o = a.G; o < 64; o <<= 1) s[e][v] += __shfl_xor(s[e][v], o, 64);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        if (lane < a.G)
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int v = 0; v < NV; ++v) red[((e * NV + v) * 4 + wave) * a.G + lane] = s[e][v];      // lane-linear: consecutive lanes, consecutive doubles (the [lane][e][v] order was a G-way bank conflict per store)
        __syncthreads();
        if ((int)threadIdx.x < a.G)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = g * 8 + e;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    double t = 0.0;
                    for (int w = 0; w < 4; ++w) t += red[((e * NV + v) * 4 + w) * a.G + g];
                    if (c < a.C) a.partial[((int64_t)blockIdx.x * a.C + c) * NV + v] = t;
                }
            }
        return;
    }
    // other group counts (12, 24, 48, 96: ngf = 96): slot p = 0 of each group adds the slots p = 1 .. PB - 1 one after the other through LDS (red: [G * 8][NV])
    for (int q = 1; q < PB; ++q) {
        __syncthreads();
        if (p == q)
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int v = 0; v < NV; ++v) red[(g * 8 + e) * NV + v] = s[e][v];
        __syncthreads();
        if (p == 0)
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int v = 0; v < NV; ++v) s[e][v] += red[(g * 8 + e) * NV + v];
    }
    if (p == 0)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = g * 8 + e;
            if (c < a.C)
#pragma unroll
                for (int v = 0; v < NV; ++v) a.partial[((int64_t)blockIdx.x * a.C + c) * NV + v] = s[e][v];
        }
}

__global__ __launch_bounds__(256) void cl_bn_stats_kernel(const ClBnArgs a) {
    extern __shared__ double red[];
    const int PB = a.pb, g = threadIdx.x % a.G, p = threadIdx.x / a.G;
    double s[8][2];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e][0] = 0.0; s[e][1] = 0.0; }
    if (p < PB) {
        float f1[8], f2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { f1[e] = 0.f; f2[e] = 0.f; }
        int run = 0;
        const int64_t stride = (int64_t)gridDim.x * PB;
        for (int64_t q = (int64_t)blockIdx.x * PB + p; q < a.P; q += 4 * stride) {
            u32x4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {      // four independent 16-byte loads in flight per thread
                const int64_t qq = q + u * stride;
                const bool ok = qq < a.P;
                const uint32_t n = fdiv((uint32_t)(ok ? qq : q), a.div_pix), pix = (uint32_t)(ok ? qq : q) - n * a.div_pix.div;
                w[u] = cl_ld(a.x + (int64_t)n * a.xv.sn + (int64_t)pix * a.xv.pitch + g * 8);
                if (!ok) w[u] = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[8];
                cl_unpack(w[u], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { f1[e] += v[e]; f2[e] += v[e] * v[e]; }
            }
            if (++run == 16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e][0] += f1[e]; s[e][1] += f2[e]; f1[e] = 0.f; f2[e] = 0.f; }
                run = 0;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e][0] += f1[e]; s[e][1] += f2[e]; }
    }
    cl_bn_block_out<2>(s, a, g, p, PB, red);
}

// sum of the partials of one channel and one value: fixed order (thread t sums blocks t, t + 256, ...; then a fixed tree over the 256 threads)
__device__ __forceinline__ double cl_partial_sum(const double* __restrict__ partial, int nblocks, int C, int c, int v, int NV, double* red) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[((int64_t)b * C + c) * NV + v];
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
// mean / invstd / running statistics from the partials (training) — one 256-thread workgroup per channel, fp64, fixed order
__global__ __launch_bounds__(256) void cl_bn_finalize_kernel(const double* __restrict__ partial, int nblocks, int C, double count, float eps, float momentum,
                                                             float* __restrict__ rm, float* __restrict__ rv, int64_t* __restrict__ nbt,
                                                             float* __restrict__ mean, float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double red[256];
    const int c = blockIdx.x;
    const double s1 = cl_partial_sum(partial, nblocks, C, c, 0, 2, red), s2 = cl_partial_sum(partial, nblocks, C, c, 1, 2, red);
    if (threadIdx.x != 0) return;
    if (c == 0 && nbt) *nbt += 1;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    invstd[c] = is;
    scale[c] = gamma[c] * is;                       // the apply pass's coefficients: y = act(mask * (x * scale + shift))
    shift[c] = beta[c] - (float)m * gamma[c] * is;
    if (rm) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * m);
        rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * unb);
    }
}
// scale / shift for the apply pass (training: batch statistics; eval: running statistics)
__global__ void cl_bn_coeff_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                   const float* __restrict__ invstd, const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float m, is;
    if (mean) { m = mean[c]; is = invstd[c]; }
    else { m = rm[c]; is = 1.f / sqrtf(rv[c] + eps); save_mean[c] = m; save_invstd[c] = is; }
    scale[c] = gamma[c] * is;
    shift[c] = beta[c] - m * gamma[c] * is;
}

// thread (p, g) of a block: 8 channels of group g, pixels p + PB (blockIdx.x + k gridDim.x): per-channel parameters live in registers
__global__ __launch_bounds__(256) void cl_bn_apply_kernel(const ClBnArgs a, int64_t total) {
    (void)total;
    const int PB = a.pb, g = threadIdx.x % a.G, p = threadIdx.x / a.G;
    if (p >= PB) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = a.scale[g * 8 + e]; sh[e] = a.shift[g * 8 + e]; }
    const bool leaky = a.act == DCV_ACT_LEAKY;
    const float slope = a.slope;
    const int64_t stride = (int64_t)gridDim.x * PB;
    for (int64_t q = (int64_t)blockIdx.x * PB + p; q < a.P; q += 2 * stride) {
        const int64_t q1 = q + stride;
        const bool two = q1 < a.P;
        const uint32_t n0 = fdiv((uint32_t)q, a.div_pix), px0 = (uint32_t)q - n0 * a.div_pix.div;
        const uint32_t n1 = two ? fdiv((uint32_t)q1, a.div_pix) : n0, px1 = two ? (uint32_t)q1 - n1 * a.div_pix.div : px0;
        const u32x4 w0 = cl_ld(a.x + (int64_t)n0 * a.xv.sn + (int64_t)px0 * a.xv.pitch + g * 8);
        const u32x4 w1 = cl_ld(a.x + (int64_t)n1 * a.xv.sn + (int64_t)px1 * a.xv.pitch + g * 8);
        float v0[8], v1[8];
        cl_unpack(w0, v0); cl_unpack(w1, v1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float z0 = v0[e] * sc[e] + sh[e], z1 = v1[e] * sc[e] + sh[e];
            if (a.mask) { z0 *= a.mask[(int64_t)n0 * a.C + g * 8 + e]; z1 *= a.mask[(int64_t)n1 * a.C + g * 8 + e]; }
            if (leaky) { z0 = z0 > 0.f ? z0 : z0 * slope; z1 = z1 > 0.f ? z1 : z1 * slope; }
            v0[e] = z0; v1[e] = z1;
        }
        cl_st(a.y + (int64_t)n0 * a.yv.sn + (int64_t)px0 * a.yv.pitch + g * 8, cl_pack(v0));
        if (two) cl_st(a.y + (int64_t)n1 * a.yv.sn + (int64_t)px1 * a.yv.pitch + g * 8, cl_pack(v1));
    }
}

// backward pass 1: per channel sum(dz), sum(dz * xhat) with dz = dy * act'(z) * mask, z = mask * (x scale + shift), xhat = (x - mean) invstd
// (c1 = mean, c2 = invstd here)
__global__ __launch_bounds__(256) void cl_bn_bwd_reduce_kernel(const ClBnArgs a) {
    extern __shared__ double red[];
    const int PB = a.pb, g = threadIdx.x % a.G, p = threadIdx.x / a.G;
    double s[8][2];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e][0] = 0.0; s[e][1] = 0.0; }
    if (p < PB) {
        float f1[8], f2[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = min(g * 8 + e, a.C - 1);
            f1[e] = 0.f; f2[e] = 0.f; mu[e] = a.mean[c]; is[e] = a.invstd[c]; sc[e] = a.gamma[c] * is[e]; sh[e] = a.beta[c] - mu[e] * sc[e];
        }
        int run = 0;
        const int64_t stride = (int64_t)gridDim.x * PB;
        for (int64_t q = (int64_t)blockIdx.x * PB + p; q < a.P; q += 2 * stride) {
            u32x4 wx[2], wd[2];
            uint32_t nn[2];
            bool okk[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {      // two pixels = four independent 16-byte loads in flight per thread
                const int64_t qq = q + u * stride;
                okk[u] = qq < a.P;
                const uint32_t n = fdiv((uint32_t)(okk[u] ? qq : q), a.div_pix), pix = (uint32_t)(okk[u] ? qq : q) - n * a.div_pix.div;
                nn[u] = n;
                wx[u] = cl_ld(a.x + (int64_t)n * a.xv.sn + (int64_t)pix * a.xv.pitch + g * 8);
                wd[u] = cl_ld(a.dy + (int64_t)n * a.dyv.sn + (int64_t)pix * a.dyv.pitch + g * 8);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float x[8], d[8];
                cl_unpack(wx[u], x); cl_unpack(wd[u], d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = g * 8 + e;
                    float mk = 1.f;
                    if (a.mask && c < a.C) mk = a.mask[(int64_t)nn[u] * a.C + c];
                    const float z = mk * (x[e] * sc[e] + sh[e]);
                    float dz = okk[u] ? d[e] * mk : 0.f;
                    if (a.act == DCV_ACT_LEAKY) dz *= z > 0.f ? 1.f : a.slope;
                    f1[e] += dz; f2[e] += dz * (x[e] - mu[e]) * is[e];
                }
            }
            if (++run == 32) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e][0] += f1[e]; s[e][1] += f2[e]; f1[e] = 0.f; f2[e] = 0.f; }
                run = 0;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e][0] += f1[e]; s[e][1] += f2[e]; }
    }
    cl_bn_block_out<2>(s, a, g, p, PB, red);
}
// dgamma = sum(dz xhat), dbeta = sum(dz); coefficients of the apply pass: dx = k1 * dz - k2 - k3 * x  with
//   training: dx = gamma invstd (dz - mean(dz) - xhat mean(dz xhat)):  k1 = gamma invstd, k3 = k1 invstd mean(dz xhat), k2 = k1 mean(dz) - k3 mean
//   eval:     dx = gamma invstd dz
__global__ __launch_bounds__(256) void cl_bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblocks, int C, double count, int training,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ k1, float* __restrict__ k2, float* __restrict__ k3) {
    __shared__ double red[256];
    const int c = blockIdx.x;
    const double s1 = cl_partial_sum(partial, nblocks, C, c, 0, 2, red), s2 = cl_partial_sum(partial, nblocks, C, c, 1, 2, red);
    if (threadIdx.x != 0) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    const double a1 = (double)gamma[c] * invstd[c];
    if (training) {
        const double a3 = a1 * invstd[c] * (s2 / count);
        k1[c] = (float)a1; k3[c] = (float)a3; k2[c] = (float)(a1 * (s1 / count) - a3 * mean[c]);
    } else {
        k1[c] = (float)a1; k3[c] = 0.f; k2[c] = 0.f;
    }
}
__global__ __launch_bounds__(256) void cl_bn_bwd_apply_kernel(const ClBnArgs a, int64_t total) {
    (void)total;
    const int PB = a.pb, g = threadIdx.x % a.G, p = threadIdx.x / a.G;
    if (p >= PB) return;
    float sc[8], sh[8], k1[8], k2[8], k3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 8 + e;
        sc[e] = a.gamma[c] * a.invstd[c]; sh[e] = a.beta[c] - a.mean[c] * sc[e];
        k1[e] = a.c1[c]; k2[e] = a.c2[c]; k3[e] = a.c3[c];
    }
    const bool leaky = a.act == DCV_ACT_LEAKY;
    const float slope = a.slope;
    const int64_t stride = (int64_t)gridDim.x * PB;
    for (int64_t q = (int64_t)blockIdx.x * PB + p; q < a.P; q += stride) {
        const uint32_t n = fdiv((uint32_t)q, a.div_pix), pix = (uint32_t)q - n * a.div_pix.div;
        float x[8], d[8];
        cl_unpack(cl_ld(a.x + (int64_t)n * a.xv.sn + (int64_t)pix * a.xv.pitch + g * 8), x);
        cl_unpack(cl_ld(a.dy + (int64_t)n * a.dyv.sn + (int64_t)pix * a.dyv.pitch + g * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float mk = 1.f;
            if (a.mask) mk = a.mask[(int64_t)n * a.C + g * 8 + e];
            const float z = mk * (x[e] * sc[e] + sh[e]);
            float dz = d[e] * mk;
            if (leaky) dz *= z > 0.f ? 1.f : slope;
            d[e] = k1[e] * dz - k2[e] - k3[e] * x[e];
        }
        cl_st(a.y + (int64_t)n * a.yv.sn + (int64_t)pix * a.yv.pitch + g * 8, cl_pack(d));
    }
}

// sample-linear view of a channels-last dims5 (pixel = (d, h, w) linear); false when the tensor is not of that form
static bool cl_view(const dcv_dims5& d, ClView* v) {
    if (d.c > 1 && d.sc != 1) return false;
    int64_t pitch = 0;
    if (d.w > 1) pitch = d.sw;
    else if (d.h > 1) pitch = d.sh;
    else if (d.d > 1) pitch = d.sd;
    else pitch = d.n > 1 ? d.sn : (d.c + 7) / 8 * 8;
    if (d.w > 1 && d.h > 1 && d.sh != (int64_t)d.w * d.sw) return false;
    if (d.h > 1 && d.d > 1 && d.sd != (int64_t)d.h * (d.w > 1 ? d.w * d.sw : d.sh) / 1) {
        if (d.sd != (int64_t)d.h * d.sh) return false;
    }
    if (d.w > 1 && d.h == 1 && d.d > 1 && d.sd != (int64_t)d.w * d.sw) return false;
    if (pitch % 8 || pitch < (d.c + 7) / 8 * 8) return false;
    v->pitch = (int32_t)pitch;
    v->sn = d.n > 1 ? d.sn : (int64_t)d.d * d.h * d.w * pitch;
    if (v->sn % 8) return false;
    return true;
}
static ClShape cl_shape(const dcv_dims5& d) {
    ClShape s;
    s.N = d.n; s.C = d.c; s.PIX = d.d * d.h * d.w; s.G = (d.c + 7) / 8;
    s.div_g = make_fastdiv((uint32_t)s.G);
    s.div_pix = make_fastdiv((uint32_t)s.PIX);
    return s;
}

#ifdef DCV_CL_FP16
}  // inline namespace clf16
#endif
}  // namespace dcv

using namespace dcv;

extern "C" {

static int cl_cvt_args(const dcv_dims5* fd, const dcv_dims5* bd, ClCvtArgs* a, const char* tag) {
    if (!fd || !bd) return fail(DCV_EINVAL, "%s: null descriptor", tag);
    if (fd->n != bd->n || fd->c != bd->c || fd->d != bd->d || fd->h != bd->h || fd->w != bd->w) return fail(DCV_EINVAL, "%s: shapes differ", tag);
    ClView v;
    if (!cl_view(*bd, &v)) return fail(DCV_EINVAL, "%s: the bf16 tensor is not sample-linear channels-last", tag);
    memset(a, 0, sizeof(*a));
    a->f_sn = fd->sn; a->f_sc = fd->sc; a->f_sd = (int32_t)fd->sd; a->f_sh = (int32_t)fd->sh; a->f_sw = (int32_t)fd->sw;
    a->D = fd->d; a->H = fd->h; a->W = fd->w; a->C = fd->c; a->Cpad = (fd->c + 7) / 8 * 8;
    a->b_sn = v.sn; a->b_pitch = v.pitch;
    const int G = a->Cpad / 8, PIX = fd->d * fd->h * fd->w;
    a->total = (int64_t)fd->n * PIX * G;
    a->div_g = make_fastdiv((uint32_t)G); a->div_pix = make_fastdiv((uint32_t)PIX);
    a->div_hw = make_fastdiv((uint32_t)(fd->h * fd->w)); a->div_w = make_fastdiv((uint32_t)fd->w);
    return DCV_OK;
}

int dcv_cl_from_f32(const float* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, void* stream) {
    ClCvtArgs a;
    int rc = cl_cvt_args(xd, yd, &a, "cl_from_f32");
    if (rc != DCV_OK) return rc;
    if (!x || !y) return fail(DCV_EINVAL, "cl_from_f32: null pointer");
    a.f = x; a.b = static_cast<cl_h*>(y);
    if (a.total == 0) return DCV_OK;
    hipLaunchKernelGGL(cl_from_f32_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}
int dcv_cl_to_f32(const void* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, int accumulate, void* stream) {
    ClCvtArgs a;
    int rc = cl_cvt_args(yd, xd, &a, "cl_to_f32");
    if (rc != DCV_OK) return rc;
    if (!x || !y) return fail(DCV_EINVAL, "cl_to_f32: null pointer");
    a.f = y; a.b = const_cast<cl_h*>(static_cast<const cl_h*>(x));
    if (a.total == 0) return DCV_OK;
    hipLaunchKernelGGL(cl_to_f32_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a, accumulate);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_cl_elementwise(int kind, const void* x, const dcv_dims5* xd, const void* z, const dcv_dims5* zd, void* y, const dcv_dims5* yd,
                       float a_, float b_, uint64_t seed, uint64_t offset, void* stream) {
    if (!x || !xd || !y || !yd) return fail(DCV_EINVAL, "cl_elementwise: null pointer");
    if (kind < 0 || kind > 6) return fail(DCV_EINVAL, "cl_elementwise: kind");
    const bool two = kind == 1 || kind == 3 || kind == 4;
    if (two && (!z || !zd)) return fail(DCV_EINVAL, "cl_elementwise: second operand missing");
    ClEwArgs a;
    memset(&a, 0, sizeof(a));
    if (!same_shape(*xd, *yd) || (two && !same_shape(*xd, *zd))) return fail(DCV_EINVAL, "cl_elementwise: shapes differ");
    if (!cl_view(*xd, &a.xv) || !cl_view(*yd, &a.yv) || (two && !cl_view(*zd, &a.zv))) return fail(DCV_EINVAL, "cl_elementwise: operands must be sample-linear channels-last");
    a.x = static_cast<const cl_h*>(x); a.z = static_cast<const cl_h*>(z); a.y = static_cast<cl_h*>(y);
    a.s = cl_shape(*xd);
    a.total = (int64_t)a.s.N * a.s.PIX * a.s.G;
    a.kind = kind; a.a = a_; a.b = b_; a.seed = seed; a.offset = offset;
    if ((int64_t)a.s.N * a.s.PIX >= (1ll << 31)) return fail(DCV_EUNSUPPORTED, "cl_elementwise: too many pixels");
    if (a.total == 0) return DCV_OK;
    hipLaunchKernelGGL(cl_ew_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

// the same from the producing convolution's per-tile fp32 sums stat[part][pitch][2] (dcv_cl_conv_forward_stats): fp64 combine, fixed order
__global__ __launch_bounds__(256) void cl_bn_finalize_stat_kernel(const float* __restrict__ stat, int nparts, int pitch, int C, double count, float eps, float momentum,
                                                                  float* __restrict__ rm, float* __restrict__ rv, int64_t* __restrict__ nbt,
                                                                  float* __restrict__ mean, float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double red[2][256];
    const int c = blockIdx.x;
    double a1 = 0.0, a2 = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 256) {
        const float2 v = *reinterpret_cast<const float2*>(stat + ((int64_t)p * pitch + c) * 2);
        a1 += (double)v.x; a2 += (double)v.y;
    }
    red[0][threadIdx.x] = a1; red[1][threadIdx.x] = a2;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const double s1 = red[0][0], s2 = red[1][0];
    if (c == 0 && nbt) *nbt += 1;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    invstd[c] = is;
    scale[c] = gamma[c] * is;
    shift[c] = beta[c] - (float)m * gamma[c] * is;
    if (rm) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * m);
        rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * unb);
    }
}

// workspace: partials (doubles) + 5 C floats of coefficients
size_t dcv_cl_bn_workspace_bytes(int channels) { return (size_t)1024 * channels * 2 * sizeof(double) + (size_t)8 * channels * sizeof(float) + 512; }

static int cl_bn_setup(const dcv_dims5* xd, ClBnArgs* a, int* blocks, const char* tag) {
    if (!xd) return fail(DCV_EINVAL, "%s: null descriptor", tag);
    if (xd->c % 8 || xd->c > 1024) return fail(DCV_EUNSUPPORTED, "%s: channel count must be a multiple of 8 (<= 1024)", tag);
    memset(a, 0, sizeof(*a));
    if (!cl_view(*xd, &a->xv)) return fail(DCV_EINVAL, "%s: input must be sample-linear channels-last", tag);
    a->N = xd->n; a->C = xd->c; a->PIX = xd->d * xd->h * xd->w; a->G = xd->c / 8;
    a->div_pix = make_fastdiv((uint32_t)a->PIX);
    a->P = (int64_t)a->N * a->PIX;
    if (a->P >= (1ll << 31)) return fail(DCV_EUNSUPPORTED, "%s: too many pixels", tag);
    a->pb = std::max(1, 256 / a->G);
    int64_t nb = (a->P + (int64_t)a->pb * 8 - 1) / ((int64_t)a->pb * 8);     // >= 8 pixels per thread
    *blocks = (int)std::min<int64_t>(std::max<int64_t>(nb, 1), 1024);
    return DCV_OK;
}

static int cl_bn_forward(const void* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd,
                         const float* mask, int training, float momentum, float eps, int act, float slope, void* ws, size_t ws_bytes, void* stream,
                         const float* stat, int nparts, int pitch);

int dcv_cl_bn_act_forward(const void* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd,
                          const float* mask, int training, float momentum, float eps, int act, float slope, void* ws, size_t ws_bytes, void* stream) {
    return cl_bn_forward(x, xd, y, yd, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_invstd, mask, training, momentum, eps, act, slope,
                         ws, ws_bytes, stream, nullptr, 0, 0);
}
// training-mode BatchNorm whose statistics come from the producing convolution's epilogue (dcv_cl_conv_forward_stats: `nparts` rows of `pitch` channels x {sum, sum^2}):
// the pass over x that cl_bn_stats_kernel makes is skipped
int dcv_cl_bn_act_forward_stats(const void* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd,
                                const float* mask, float momentum, float eps, int act, float slope, const float* stat, int nparts, int pitch,
                                void* ws, size_t ws_bytes, void* stream) {
    if (!stat || nparts <= 0 || !xd || pitch < xd->c) return fail(DCV_EINVAL, "cl_bn_act_forward_stats: partial sums missing or narrower than the channel count");
    return cl_bn_forward(x, xd, y, yd, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_invstd, mask, 1, momentum, eps, act, slope,
                         ws, ws_bytes, stream, stat, nparts, pitch);
}

static int cl_bn_forward(const void* x, const dcv_dims5* xd, void* y, const dcv_dims5* yd, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd,
                         const float* mask, int training, float momentum, float eps, int act, float slope, void* ws, size_t ws_bytes, void* stream,
                         const float* stat, int nparts, int pitch) {
    ClBnArgs a;
    int blocks = 0;
    int rc = cl_bn_setup(xd, &a, &blocks, "cl_bn_act_forward");
    if (rc != DCV_OK) return rc;
    if (!x || !y || !yd || !gamma || !beta || !save_mean || !save_invstd || !ws) return fail(DCV_EINVAL, "cl_bn_act_forward: null pointer");
    if (!same_shape(*xd, *yd) || !cl_view(*yd, &a.yv)) return fail(DCV_EINVAL, "cl_bn_act_forward: output must be sample-linear channels-last of the input's shape");
    if (ws_bytes < dcv_cl_bn_workspace_bytes(a.C)) return fail(DCV_EWORKSPACE, "cl_bn_act_forward: workspace too small");
    if (act != DCV_ACT_NONE && act != DCV_ACT_LEAKY) return fail(DCV_EUNSUPPORTED, "cl_bn_act_forward: activation");
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* partial = static_cast<double*>(ws);
    float* coef = reinterpret_cast<float*>(static_cast<char*>(ws) + (size_t)1024 * a.C * 2 * sizeof(double));
    a.x = static_cast<const cl_h*>(x); a.y = static_cast<cl_h*>(y); a.partial = partial; a.mask = mask; a.act = act; a.slope = slope;
    a.scale = coef; a.shift = coef + a.C;
    const int cb = (a.C + 63) / 64;
    if (training && stat) {
        hipLaunchKernelGGL(cl_bn_finalize_stat_kernel, dim3((unsigned)a.C), dim3(256), 0, st, stat, nparts, pitch, a.C, (double)a.P, eps, momentum, running_mean, running_var,
                           num_batches_tracked, save_mean, save_invstd, gamma, beta, coef, coef + a.C);
    } else if (training) {
        hipLaunchKernelGGL(cl_bn_stats_kernel, dim3((unsigned)blocks), dim3(256), (size_t)4 * a.C * 2 * sizeof(double), st, a);
        DCV_LAUNCH_CHECK();
        hipLaunchKernelGGL(cl_bn_finalize_kernel, dim3((unsigned)a.C), dim3(256), 0, st, partial, blocks, a.C, (double)a.P, eps, momentum, running_mean, running_var,
                           num_batches_tracked, save_mean, save_invstd, gamma, beta, coef, coef + a.C);
    } else {
        if (!running_mean || !running_var) return fail(DCV_EINVAL, "cl_bn_act_forward: eval mode needs running statistics");
        hipLaunchKernelGGL(cl_bn_coeff_kernel, dim3((unsigned)cb), dim3(64), 0, st, a.C, gamma, beta, (const float*)nullptr, (const float*)nullptr, running_mean, running_var, eps,
                           coef, coef + a.C, save_mean, save_invstd);
    }
    DCV_LAUNCH_CHECK();
    const int64_t total = a.P * a.G;
    hipLaunchKernelGGL(cl_bn_apply_kernel, dim3((unsigned)std::min<int64_t>(std::max<int64_t>((a.P + (int64_t)a.pb * 4 - 1) / ((int64_t)a.pb * 4), 1), 8192)), dim3(256), 0, st, a, total);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_cl_bn_act_backward(const void* dy, const dcv_dims5* dyd, const void* x, const dcv_dims5* xd, void* dx, const dcv_dims5* dxd,
                           const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, const float* mask,
                           int training, int act, float slope, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    ClBnArgs a;
    int blocks = 0;
    int rc = cl_bn_setup(xd, &a, &blocks, "cl_bn_act_backward");
    if (rc != DCV_OK) return rc;
    if (!dy || !dyd || !x || !dx || !dxd || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta || !ws) return fail(DCV_EINVAL, "cl_bn_act_backward: null pointer");
    if (!same_shape(*xd, *dyd) || !same_shape(*xd, *dxd) || !cl_view(*dyd, &a.dyv) || !cl_view(*dxd, &a.yv))
        return fail(DCV_EINVAL, "cl_bn_act_backward: operands must be sample-linear channels-last of one shape");
    if (ws_bytes < dcv_cl_bn_workspace_bytes(a.C)) return fail(DCV_EWORKSPACE, "cl_bn_act_backward: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* partial = static_cast<double*>(ws);
    float* coef = reinterpret_cast<float*>(static_cast<char*>(ws) + (size_t)1024 * a.C * 2 * sizeof(double));
    a.x = static_cast<const cl_h*>(x); a.dy = static_cast<const cl_h*>(dy); a.y = static_cast<cl_h*>(dx); a.partial = partial; a.mask = mask; a.act = act; a.slope = slope;
    a.scale = coef; a.shift = coef + a.C;
    const int cb = (a.C + 63) / 64;
    a.gamma = gamma; a.beta = beta; a.mean = save_mean; a.invstd = save_invstd;
    a.c1 = save_mean; a.c2 = save_invstd;
    hipLaunchKernelGGL(cl_bn_bwd_reduce_kernel, dim3((unsigned)blocks), dim3(256), (size_t)4 * a.C * 2 * sizeof(double), st, a);
    DCV_LAUNCH_CHECK();
    float* k1 = coef + 2 * a.C; float* k2 = coef + 3 * a.C; float* k3 = coef + 4 * a.C;
    hipLaunchKernelGGL(cl_bn_bwd_finalize_kernel, dim3((unsigned)a.C), dim3(256), 0, st, partial, blocks, a.C, (double)a.P, training, gamma, save_mean, save_invstd,
                       dgamma, dbeta, k1, k2, k3);
    DCV_LAUNCH_CHECK();
    a.c1 = k1; a.c2 = k2; a.c3 = k3;
    const int64_t total = a.P * a.G;
    hipLaunchKernelGGL(cl_bn_bwd_apply_kernel, dim3((unsigned)std::min<int64_t>(std::max<int64_t>((a.P + (int64_t)a.pb * 4 - 1) / ((int64_t)a.pb * 4), 1), 8192)), dim3(256), 0, st, a, total);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

}  // extern "C"
