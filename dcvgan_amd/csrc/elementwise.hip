// HBM-bound kernels of the DCVGAN step for gfx950: BatchNorm (+Dropout2d +activation)
// forward/backward, activations, axpby / strided copy, Philox noise, GAN losses,
// GRU recurrence, Adam.  All work on NCDHW views: a "row" is one (n, c, d) plane
// of H*W contiguous floats (vectorised 16 B per lane); views whose (h, w) plane is
// not contiguous fall back to rows of one element.
#include "dcv_common.h"
#include <algorithm>

namespace dcv {

struct RowView {
    int64_t sn, sc, sd, sh, sw;
};

// rows = N*C*D*(H*W/inner); inner = contiguous run (H*W or 1)
struct RowMap {
    int32_t N, C, D, H, W;
    int32_t inner;   // contiguous run length in elements
    int32_t vec;     // 4 if inner % 4 == 0 and all views 16-B friendly, else 1
    int32_t gpr;     // groups (of vec elements) per row
    FastDiv div_gpr, div_cd, div_d, div_w;  // g -> row, row -> n, (c,d); tail -> (h,w) when inner == 1
    int64_t groups;
};

static bool plane_contig(const dcv_dims5& v) { return (v.w == 1 || v.sw == 1) && (v.h == 1 || v.sh == v.w); }

static RowMap make_rowmap(const dcv_dims5& shape, const dcv_dims5* const* views, int nviews, const void* const* ptrs) {
    RowMap m;
    memset(&m, 0, sizeof(m));
    m.N = shape.n; m.C = shape.c; m.D = shape.d; m.H = shape.h; m.W = shape.w;
    bool contig = true;
    for (int i = 0; i < nviews; ++i) contig = contig && plane_contig(*views[i]);
    const int hw = shape.h * shape.w;
    m.inner = contig ? hw : 1;
    bool v4 = contig && (hw % 4 == 0);
    if (v4)
        for (int i = 0; i < nviews; ++i) {
            const dcv_dims5& v = *views[i];
            if ((v.sn % 4) || (v.sc % 4) || (v.sd % 4) || (reinterpret_cast<uintptr_t>(ptrs[i]) % 16)) v4 = false;
        }
    m.vec = v4 ? 4 : 1;
    m.gpr = contig ? hw / m.vec : hw;  // when not contiguous: one group per element, "row" = (n,c,d)
    m.div_gpr = make_fastdiv((uint32_t)m.gpr);
    m.div_cd = make_fastdiv((uint32_t)(shape.c * shape.d));
    m.div_d = make_fastdiv((uint32_t)shape.d);
    m.div_w = make_fastdiv((uint32_t)shape.w);
    m.groups = (int64_t)shape.n * shape.c * shape.d * m.gpr;
    return m;
}

struct Pos {
    int n, c, d, col;  // col: element offset inside the (h, w) plane (first of the group)
    int h, w;
};

__device__ __forceinline__ Pos locate(const RowMap& m, uint32_t g) {
    Pos p;
    const uint32_t row = fdiv(g, m.div_gpr);
    const uint32_t cg = g - row * m.div_gpr.div;
    const uint32_t n = fdiv(row, m.div_cd);
    const uint32_t cd = row - n * m.div_cd.div;
    const uint32_t c = fdiv(cd, m.div_d);
    p.n = (int)n; p.c = (int)c; p.d = (int)(cd - c * m.div_d.div);
    p.col = (int)cg * m.vec;
    if (m.inner == 1) {
        const uint32_t h = fdiv(cg, m.div_w);
        p.h = (int)h; p.w = (int)(cg - h * m.div_w.div);
    } else {
        p.h = 0; p.w = 0;
    }
    return p;
}

__device__ __forceinline__ int64_t offs(const RowMap& m, const RowView& v, const Pos& p) {
    int64_t o = (int64_t)p.n * v.sn + (int64_t)p.c * v.sc + (int64_t)p.d * v.sd;
    if (m.inner == 1) o += (int64_t)p.h * v.sh + (int64_t)p.w * v.sw;
    else o += p.col;
    return o;
}

static RowView rv(const dcv_dims5& d) { return RowView{d.sn, d.sc, d.sd, d.sh, d.sw}; }

__device__ __forceinline__ float act_fwd(float v, int act, float slope) {
    if (act == DCV_ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (act == DCV_ACT_TANH) return tanhf(v);
    return v;
}
// derivative evaluated from the OUTPUT y
__device__ __forceinline__ float act_bwd_from_y(float y, int act, float slope) {
    if (act == DCV_ACT_LEAKY) return y > 0.f ? 1.f : slope;
    if (act == DCV_ACT_TANH) return 1.f - y * y;
    return 1.f;
}

static inline unsigned grid_for(int64_t groups) {
    int64_t b = (groups + 255) / 256;
    if (b > 8192) b = 8192;   // 256 CUs x 8 blocks x 4: grid-stride beyond
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ------------------------------------------------------------------------- //
// generic unary / binary elementwise kernels
// ------------------------------------------------------------------------- //
template <int VEC, class F>
__global__ __launch_bounds__(256) void ew_kernel(RowMap m, F f) {
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < m.groups; g += (int64_t)gridDim.x * 256) {
        const Pos p = locate(m, (uint32_t)g);
        f.template apply<VEC>(m, p);
    }
}

// Rows (one (n, c, d) plane = H * W contiguous floats) of at least 1024 floats: one workgroup per row.  Sample, channel and depth are
// then workgroup-uniform — the per-channel parameters arrive as scalar loads and the address is one scalar base plus the thread's
// column — instead of three magic divisions, 64-bit stride arithmetic and gathered parameter loads per 16 bytes of payload, which
// kept the generic kernel at 4.8 TB/s on a 1.17 GB activation where torch's plain elementwise kernel streams at 6.3.
template <class F>
__global__ __launch_bounds__(256) void ew_rows_kernel(RowMap m, F f) {
    const uint32_t row = blockIdx.x;
    Pos p;
    const uint32_t n = fdiv(row, m.div_cd);
    const uint32_t cd = row - n * m.div_cd.div;
    const uint32_t c = fdiv(cd, m.div_d);
    p.n = (int)n; p.c = (int)c; p.d = (int)(cd - c * m.div_d.div);
    p.h = 0; p.w = 0;
    for (int g = threadIdx.x; g < m.gpr; g += 256) {
        p.col = g * 4;
        f.template apply<4>(m, p);
    }
}

static bool ew_rows_off() {
    static const bool off = getenv("DCV_NO_EW_ROWS") != nullptr;
    return off;
}

template <class F>
static int launch_ew(const RowMap& m, const F& f, hipStream_t s) {
    if (m.groups >= (1ll << 32)) return fail(DCV_EUNSUPPORTED, "elementwise: tensor too large");
    if (m.groups == 0) return DCV_OK;
    const int64_t rows = m.groups / m.gpr;
    if (m.vec == 4 && m.inner != 1 && m.gpr % 256 == 0 && rows < (1ll << 31) && !ew_rows_off())
        hipLaunchKernelGGL((ew_rows_kernel<F>), dim3((unsigned)rows), dim3(256), 0, s, m, f);
    else if (m.vec == 4) hipLaunchKernelGGL((ew_kernel<4, F>), dim3(grid_for(m.groups)), dim3(256), 0, s, m, f);
    else hipLaunchKernelGGL((ew_kernel<1, F>), dim3(grid_for(m.groups)), dim3(256), 0, s, m, f);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

template <int VEC> struct Vec;
template <> struct Vec<4> { typedef float4 T; };
template <> struct Vec<1> { typedef float T; };
typedef float ew_f4 __attribute__((ext_vector_type(4)));
// DCV_EW_NT (A/B builds): non-temporal loads / stores in the streaming kernels (every byte is touched once per pass)
#ifdef DCV_EW_NT
#define DCV_EW_LD4(P) __builtin_nontemporal_load(reinterpret_cast<const ew_f4*>(P))
#define DCV_EW_ST4(P, V) __builtin_nontemporal_store((V), reinterpret_cast<ew_f4*>(P))
#else
#define DCV_EW_LD4(P) (*reinterpret_cast<const ew_f4*>(P))
#define DCV_EW_ST4(P, V) (*reinterpret_cast<ew_f4*>(P) = (V))
#endif
template <int VEC> __device__ __forceinline__ void ld(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) { const ew_f4 t = DCV_EW_LD4(p); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    else v[0] = *p;
}
template <int VEC> __device__ __forceinline__ void st(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) { const ew_f4 t = {v[0], v[1], v[2], v[3]}; DCV_EW_ST4(p, t); }
    else *p = v[0];
}

struct ActFwd {
    const float* x; float* y; RowView xv, yv; int act; float slope;
    template <int VEC> __device__ void apply(const RowMap& m, const Pos& p) const {
        float v[VEC];
        ld<VEC>(x + offs(m, xv, p), v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = act_fwd(v[i], act, slope);
        st<VEC>(y + offs(m, yv, p), v);
    }
};
struct ActBwd {
    const float* dy; const float* y; float* dx; RowView dyv, yv, dxv; int act; float slope;
    template <int VEC> __device__ void apply(const RowMap& m, const Pos& p) const {
        float g[VEC], o[VEC];
        ld<VEC>(dy + offs(m, dyv, p), g);
        ld<VEC>(y + offs(m, yv, p), o);
#pragma unroll
        for (int i = 0; i < VEC; ++i) g[i] *= act_bwd_from_y(o[i], act, slope);
        st<VEC>(dx + offs(m, dxv, p), g);
    }
};
struct Axpby {
    const float* x; const float* z; float* y; RowView xv, zv, yv; float a, b;
    template <int VEC> __device__ void apply(const RowMap& m, const Pos& p) const {
        float u[VEC];
        ld<VEC>(x + offs(m, xv, p), u);
        if (z) {
            float w[VEC];
            ld<VEC>(z + offs(m, zv, p), w);
#pragma unroll
            for (int i = 0; i < VEC; ++i) u[i] = a * u[i] + b * w[i];
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) u[i] = a * u[i];
        }
        st<VEC>(y + offs(m, yv, p), u);
    }
};

// ------------------------------------------------------------------------- //
// Philox4x32-10 + Box-Muller
// ------------------------------------------------------------------------- //
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
// 4 N(0,1) samples for 128-bit counter (idx, stream offset)
__device__ __forceinline__ void normal4(uint64_t seed, uint64_t offset, uint64_t idx, float (&o)[4]) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float s = 2.3283064365386963e-10f;  // 2^-32
    const float u0 = ((float)c[0] + 0.5f) * s, u1 = ((float)c[1] + 0.5f) * s;
    const float u2 = ((float)c[2] + 0.5f) * s, u3 = ((float)c[3] + 0.5f) * s;
    const float r0 = sqrtf(-2.f * __logf(fmaxf(u0, 1e-30f))), r1 = sqrtf(-2.f * __logf(fmaxf(u2, 1e-30f)));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    o[0] = r0 * c0; o[1] = r0 * s0; o[2] = r1 * c1; o[3] = r1 * s1;
}

struct NoiseAdd {
    const float* x; float* y; RowView xv, yv; float sigma; uint64_t seed, offset;
    template <int VEC> __device__ void apply(const RowMap& m, const Pos& p) const {
        float u[VEC];
        ld<VEC>(x + offs(m, xv, p), u);
        // logical (contiguous) element index -> counter; independent of the views' strides
        const uint64_t row = ((uint64_t)p.n * m.C + p.c) * m.D + p.d;
        const uint64_t e = row * ((uint64_t)m.H * m.W) + (m.inner == 1 ? (uint64_t)p.h * m.W + p.w : (uint64_t)p.col);
        float z[4];
        normal4(seed, offset, e >> 2, z);
        if constexpr (VEC == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] += sigma * z[i];
        } else {
            u[0] += sigma * z[e & 3];
        }
        st<VEC>(y + offs(m, yv, p), u);
    }
};

__global__ void normal_fill_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= n) return;
    float z[4];
    normal4(seed, offset, (uint64_t)q, z);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (q * 4 + i < n) out[q * 4 + i] = z[i];
}

// `count` consecutive draws of n values each in one launch: draw j is normal_fill_kernel's output for offset + j (the generator's per-frame motion noise: 16 launches of 5 000
// values each were 16 dependent launch latencies at the start of every generator pass)
__global__ void normal_fill_many_kernel(float* __restrict__ out, int64_t n, int64_t count, uint64_t seed, uint64_t offset) {
    const int64_t qpd = (n + 3) / 4, t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= qpd * count) return;
    const int64_t j = t / qpd, q = t - j * qpd;
    float z[4];
    normal4(seed, offset + (uint64_t)j, (uint64_t)q, z);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (q * 4 + i < n) out[j * n + q * 4 + i] = z[i];
}

__global__ void dropout_mask_kernel(float* __restrict__ mask, int64_t n, float p, uint64_t seed, uint64_t offset) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q * 4 >= n) return;
    uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float keep = 1.f / (1.f - p);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (q * 4 + i < n) mask[q * 4 + i] = (((float)c[i] + 0.5f) * 2.3283064365386963e-10f >= p) ? keep : 0.f;
}

// ------------------------------------------------------------------------- //
// BatchNorm
// ------------------------------------------------------------------------- //
// wave64 sum via DPP-free shuffles, then LDS across the 4 waves
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red /* [4][NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = red[i] + red[NV + i] + red[2 * NV + i] + red[3 * NV + i];
}

// channel-wise iteration: groups of channel c are g = (n*D + d)*gpr + cg
struct ChanMap {
    int32_t N, C, D, H, W, inner, vec, gpr;
    FastDiv div_gpr, div_d, div_w;
    int64_t per_chan;   // groups per channel
    int32_t split;      // blocks per channel
    int64_t seg;        // groups per block
};
static ChanMap make_chanmap(const RowMap& m) {
    ChanMap c;
    memset(&c, 0, sizeof(c));
    c.N = m.N; c.C = m.C; c.D = m.D; c.H = m.H; c.W = m.W; c.inner = m.inner; c.vec = m.vec; c.gpr = m.gpr;
    c.div_gpr = m.div_gpr; c.div_d = m.div_d; c.div_w = m.div_w;
    c.per_chan = (int64_t)m.N * m.D * m.gpr;
    int64_t split = (2048 + m.C - 1) / m.C;           // ~2048 blocks in total
    const int64_t maxsplit = (c.per_chan + 1023) / 1024;  // >= 4 groups per thread
    if (split > maxsplit) split = maxsplit;
    if (split < 1) split = 1;
    c.split = (int)split;
    c.seg = (c.per_chan + split - 1) / split;
    return c;
}
__device__ __forceinline__ Pos locate_c(const ChanMap& m, int c, uint32_t g) {
    Pos p;
    const uint32_t nd = fdiv(g, m.div_gpr);
    const uint32_t cg = g - nd * m.div_gpr.div;
    const uint32_t n = fdiv(nd, m.div_d);
    p.n = (int)n; p.c = c; p.d = (int)(nd - n * m.div_d.div);
    p.col = (int)cg * m.vec;
    if (m.inner == 1) {
        const uint32_t h = fdiv(cg, m.div_w);
        p.h = (int)h; p.w = (int)(cg - h * m.div_w.div);
    } else { p.h = 0; p.w = 0; }
    return p;
}
__device__ __forceinline__ int64_t offs_c(const ChanMap& m, const RowView& v, const Pos& p) {
    int64_t o = (int64_t)p.n * v.sn + (int64_t)p.c * v.sc + (int64_t)p.d * v.sd;
    if (m.inner == 1) o += (int64_t)p.h * v.sh + (int64_t)p.w * v.sw;
    else o += p.col;
    return o;
}

// partial[c][s] = {sum x, sum x^2} over the block's segment
template <int VEC>
__global__ __launch_bounds__(256) void bn_stats_kernel(ChanMap m, const float* __restrict__ x, RowView xv, double* __restrict__ partial) {
    __shared__ double red[8];
    const int c = blockIdx.x / m.split, s = blockIdx.x % m.split;
    const int64_t g0 = (int64_t)s * m.seg;
    int64_t g1 = g0 + m.seg;
    if (g1 > m.per_chan) g1 = m.per_chan;
    float a0 = 0.f, a1 = 0.f;
    double acc[2] = {0.0, 0.0};
    int cnt = 0;
    for (int64_t g = g0 + threadIdx.x; g < g1; g += 256) {
        const Pos p = locate_c(m, c, (uint32_t)g);
        float v[VEC];
        ld<VEC>(x + offs_c(m, xv, p), v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) { a0 += v[i]; a1 += v[i] * v[i]; }
        if (++cnt == 64) { acc[0] += a0; acc[1] += a1; a0 = a1 = 0.f; cnt = 0; }  // bound fp32 run length
    }
    acc[0] += a0; acc[1] += a1;
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) { partial[(int64_t)blockIdx.x * 2] = acc[0]; partial[(int64_t)blockIdx.x * 2 + 1] = acc[1]; }
}

__device__ __forceinline__ void bn_finalize_channel(int c, double s0, double s1, double count, float eps, float momentum,
                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var) {
    const double mean = s0 / count;
    double var = s1 / count - mean * mean;
    if (var < 0) var = 0;
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = count > 1 ? var * count / (count - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// `nbt`: the module's num_batches_tracked buffer (int64, may be NULL), bumped here instead of by a launch of its own
__global__ void bn_finalize_kernel(const double* __restrict__ partial, int C, int split, double count, float eps, float momentum,
                                   float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, int64_t* __restrict__ nbt) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    double s0 = 0, s1 = 0;
    for (int k = 0; k < split; ++k) { s0 += partial[((int64_t)c * split + k) * 2]; s1 += partial[((int64_t)c * split + k) * 2 + 1]; }
    bn_finalize_channel(c, s0, s1, count, eps, momentum, save_mean, save_invstd, running_mean, running_var);
}

// conv -> BatchNorm pairs: {sum, sum^2} in fp64 from the conv epilogue's per-tile fp32 sums stat[part][pitch][2], finalised
// in the same launch (one block per channel)
__global__ __launch_bounds__(256) void bn_partials_finalize_kernel(const float* __restrict__ stat, int nparts, int pitch, double count, float eps, float momentum,
                                                                   float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                                   float* __restrict__ running_mean, float* __restrict__ running_var, int64_t* __restrict__ nbt) {
    __shared__ double red[8];
    const int c = blockIdx.x;
    double acc[2] = {0.0, 0.0};
    for (int p = threadIdx.x; p < nparts; p += 256) {
        const float2 v = *reinterpret_cast<const float2*>(stat + ((int64_t)p * pitch + c) * 2);
        acc[0] += (double)v.x;
        acc[1] += (double)v.y;
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        if (c == 0 && nbt) *nbt += 1;
        bn_finalize_channel(c, acc[0], acc[1], count, eps, momentum, save_mean, save_invstd, running_mean, running_var);
    }
}

__global__ void bn_eval_stats_kernel(int C, float eps, const float* __restrict__ rm, const float* __restrict__ rv_, float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rm[c];
    invstd[c] = 1.f / sqrtf(rv_[c] + eps);
}

struct BnApply {
    const float* x; float* y; RowView xv, yv;
    const float* gamma; const float* beta; const float* mean; const float* invstd; const float* mask;
    int act; float slope;
    template <int VEC> __device__ void apply(const RowMap& m, const Pos& p) const {
        float v[VEC];
        ld<VEC>(x + offs(m, xv, p), v);
        const float sc = gamma[p.c] * invstd[p.c];
        const float sh = beta[p.c] - mean[p.c] * sc;
        const float mk = mask ? mask[(int64_t)p.n * m.C + p.c] : 1.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = act_fwd((v[i] * sc + sh) * mk, act, slope);
        st<VEC>(y + offs(m, yv, p), v);
    }
};

// backward pass 1: partial[c][s] = {sum dz, sum dz * xhat},  dz = dy * act'(z) * mask
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(ChanMap m, const float* __restrict__ dy, RowView dyv, const float* __restrict__ x, RowView xv,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ mask,
                                                            int act, float slope, double* __restrict__ partial) {
    __shared__ double red[8];
    const int c = blockIdx.x / m.split, s = blockIdx.x % m.split;
    const int64_t g0 = (int64_t)s * m.seg;
    int64_t g1 = g0 + m.seg;
    if (g1 > m.per_chan) g1 = m.per_chan;
    const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    // the activation's branch is decided by EXACTLY the expression the forward pass evaluated (BnApply): a
    // pre-activation within rounding of zero must not come out positive there and non-positive here
    const float sc = ga * is, sh = be - mu * sc;
    float a0 = 0.f, a1 = 0.f;
    double acc[2] = {0.0, 0.0};
    int cnt = 0;
    for (int64_t g = g0 + threadIdx.x; g < g1; g += 256) {
        const Pos p = locate_c(m, c, (uint32_t)g);
        float v[VEC], d[VEC];
        ld<VEC>(x + offs_c(m, xv, p), v);
        ld<VEC>(dy + offs_c(m, dyv, p), d);
        const float mk = mask ? mask[(int64_t)p.n * m.C + c] : 1.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float xh = (v[i] - mu) * is;
            const float z = (v[i] * sc + sh) * mk;
            float dz = d[i] * mk;
            if (act == DCV_ACT_LEAKY) dz *= (z > 0.f ? 1.f : slope);
            else if (act == DCV_ACT_TANH) { const float t = tanhf(z); dz *= 1.f - t * t; }
            a0 += dz; a1 += dz * xh;
        }
        if (++cnt == 64) { acc[0] += a0; acc[1] += a1; a0 = a1 = 0.f; cnt = 0; }
    }
    acc[0] += a0; acc[1] += a1;
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) { partial[(int64_t)blockIdx.x * 2] = acc[0]; partial[(int64_t)blockIdx.x * 2 + 1] = acc[1]; }
}

// The same sums for planes of at least 1024 contiguous floats: a workgroup takes whole (n, d) rows of its channel, so the row's base
// addresses are scalar and the inner loop is two 16-byte loads per thread and trip with nothing else to compute (the generic kernel
// decodes every group with three magic divisions); the trips of a row are unrolled, which puts up to eight loads in flight per thread.
__global__ __launch_bounds__(256) void bn_bwd_reduce_rows_kernel(ChanMap m, const float* __restrict__ dy, RowView dyv, const float* __restrict__ x, RowView xv,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ mask,
                                                                 int act, float slope, double* __restrict__ partial) {
    __shared__ double red[8];
    const int c = blockIdx.x / m.split, s = blockIdx.x % m.split;
    const int rows = m.N * m.D, rps = (rows + m.split - 1) / m.split;
    const int r0 = s * rps, r1 = min(rows, r0 + rps);
    const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    const float sc = ga * is, sh = be - mu * sc;   // the forward pass's expression decides the activation's branch (see above)
    float a0 = 0.f, a1 = 0.f;
    double acc[2] = {0.0, 0.0};
    int cnt = 0;
    for (int row = r0; row < r1; ++row) {
        const uint32_t n = fdiv((uint32_t)row, m.div_d);
        const int d = row - (int)n * m.D;
        const float* __restrict__ xr = x + (int64_t)n * xv.sn + (int64_t)c * xv.sc + (int64_t)d * xv.sd + 4 * threadIdx.x;
        const float* __restrict__ dr = dy + (int64_t)n * dyv.sn + (int64_t)c * dyv.sc + (int64_t)d * dyv.sd + 4 * threadIdx.x;
        const float mk = mask ? mask[(int64_t)n * m.C + c] : 1.f;
#pragma unroll 4
        for (int g = 0; g < m.gpr; g += 256) {
            const ew_f4 v = DCV_EW_LD4(xr + 4 * g);
            const ew_f4 dd = DCV_EW_LD4(dr + 4 * g);
            const float vv[4] = {v[0], v[1], v[2], v[3]}, dv[4] = {dd[0], dd[1], dd[2], dd[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (vv[i] - mu) * is;
                const float z = (vv[i] * sc + sh) * mk;
                float dz = dv[i] * mk;
                if (act == DCV_ACT_LEAKY) dz *= (z > 0.f ? 1.f : slope);
                else if (act == DCV_ACT_TANH) { const float t = tanhf(z); dz *= 1.f - t * t; }
                a0 += dz; a1 += dz * xh;
            }
        }
        cnt += m.gpr >> 8;
        if (cnt >= 64) { acc[0] += a0; acc[1] += a1; a0 = a1 = 0.f; cnt = 0; }   // bound the fp32 run length (between rows: the trips stay branch-free)
    }
    acc[0] += a0; acc[1] += a1;
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) { partial[(int64_t)blockIdx.x * 2] = acc[0]; partial[(int64_t)blockIdx.x * 2 + 1] = acc[1]; }
}

// dbeta = sum dz, dgamma = sum dz*xhat; coef[c] = {dbeta/count, dgamma/count} for pass 2
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ partial, int C, int split, double count,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0, s1 = 0;
    for (int k = 0; k < split; ++k) { s0 += partial[((int64_t)c * split + k) * 2]; s1 += partial[((int64_t)c * split + k) * 2 + 1]; }
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
    coef[2 * c] = (float)(s0 / count);
    coef[2 * c + 1] = (float)(s1 / count);
}

struct BnBwdApply {
    const float* dy; const float* x; float* dx; RowView dyv, xv, dxv;
    const float* gamma; const float* beta; const float* mean; const float* invstd; const float* mask; const float* coef;
    int act; float slope; int training;
    template <int VEC> __device__ void apply(const RowMap& m, const Pos& p) const {
        float v[VEC], d[VEC];
        ld<VEC>(x + offs(m, xv, p), v);
        ld<VEC>(dy + offs(m, dyv, p), d);
        const int c = p.c;
        const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
        const float mk = mask ? mask[(int64_t)p.n * m.C + c] : 1.f;
        const float k0 = training ? coef[2 * c] : 0.f, k1 = training ? coef[2 * c + 1] : 0.f;
        const float sc = ga * is, sh = be - mu * sc;   // as in BnApply: same branch of the activation as the forward pass
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float xh = (v[i] - mu) * is;
            const float z = (v[i] * sc + sh) * mk;
            float dz = d[i] * mk;
            if (act == DCV_ACT_LEAKY) dz *= (z > 0.f ? 1.f : slope);
            else if (act == DCV_ACT_TANH) { const float t = tanhf(z); dz *= 1.f - t * t; }
            v[i] = ga * is * (dz - k0 - xh * k1);
        }
        st<VEC>(dx + offs(m, dxv, p), v);
    }
};

// ------------------------------------------------------------------------- //
// Small BatchNorm layers, one launch per pass (round 3).  The 4x4 / 2x2 / 1x1-spatial layers of the generators and the
// discriminators' trunks hold <= 32 K elements per channel: their passes are latency-bound ~5-25 us launches (statistics or
// partials-finalize + apply forward; reduce + finalize + apply backward: 40 + 40 + 53 launches per iteration, ~3 ms of kernel
// time).  Here ONE workgroup of 1024 threads owns a channel, keeps its elements in registers (<= 8 groups of 4 per thread and
// tensor), reduces over the workgroup in a fixed order (fp32 per thread over <= 32 terms, fp64 across lanes and waves),
// and applies — each tensor is read once and the pass is one launch.  Same expressions as the large-tensor kernels, so the
// (Leaky)ReLU branch decided in the backward pass is the forward pass's.
// ------------------------------------------------------------------------- //
constexpr int SMALL_NT = 1024, SMALL_K = 8;

__device__ __forceinline__ void block_sum_1024(double (&v)[2], double* red /* [16][2] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v[0] = wave_sum(v[0]); v[1] = wave_sum(v[1]);
    __syncthreads();
    if (lane == 0) { red[wave * 2] = v[0]; red[wave * 2 + 1] = v[1]; }
    __syncthreads();
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int w = 0; w < SMALL_NT / 64; ++w) { a += red[w * 2]; b += red[w * 2 + 1]; }
    v[0] = a; v[1] = b;
}

__global__ __launch_bounds__(SMALL_NT) void bn_fwd_small_kernel(ChanMap m, const float* __restrict__ x, RowView xv, float* __restrict__ y, RowView yv,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                                float* __restrict__ save_mean, float* __restrict__ save_invstd, const float* __restrict__ mask,
                                                                double count, float eps, float momentum, int act, float slope) {
    __shared__ double red[2 * SMALL_NT / 64];
    const int c = blockIdx.x;
    float4 v[SMALL_K];
    int32_t oy[SMALL_K];
    float mk[SMALL_K];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < SMALL_K; ++k) {
        const int64_t g = (int64_t)threadIdx.x + (int64_t)k * SMALL_NT;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f); oy[k] = -1; mk[k] = 1.f;
        if (g < m.per_chan) {
            const Pos p = locate_c(m, c, (uint32_t)g);
            v[k] = *reinterpret_cast<const float4*>(x + offs_c(m, xv, p));
            oy[k] = (int32_t)offs_c(m, yv, p);
            if (mask) mk[k] = mask[(int64_t)p.n * m.C + c];
            a0 += (v[k].x + v[k].y) + (v[k].z + v[k].w);
            a1 += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
        }
    }
    double acc[2] = {(double)a0, (double)a1};
    block_sum_1024(acc, red);
    const double mean_d = acc[0] / count;
    double var = acc[1] / count - mean_d * mean_d;
    if (var < 0) var = 0;
    const float mu = (float)mean_d, is = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        if (c == 0 && nbt) *nbt += 1;
        bn_finalize_channel(c, acc[0], acc[1], count, eps, momentum, save_mean, save_invstd, running_mean, running_var);
    }
    const float sc = gamma[c] * is, sh = beta[c] - mu * sc;      // BnApply's expressions
#pragma unroll
    for (int k = 0; k < SMALL_K; ++k)
        if (oy[k] >= 0) {
            float4 o;
            o.x = act_fwd((v[k].x * sc + sh) * mk[k], act, slope);
            o.y = act_fwd((v[k].y * sc + sh) * mk[k], act, slope);
            o.z = act_fwd((v[k].z * sc + sh) * mk[k], act, slope);
            o.w = act_fwd((v[k].w * sc + sh) * mk[k], act, slope);
            *reinterpret_cast<float4*>(y + oy[k]) = o;
        }
}

__global__ __launch_bounds__(SMALL_NT) void bn_bwd_small_kernel(ChanMap m, const float* __restrict__ dy, RowView dyv, const float* __restrict__ x, RowView xv,
                                                                float* __restrict__ dx, RowView dxv, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ mask,
                                                                double count, int act, float slope, int training, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[2 * SMALL_NT / 64];
    const int c = blockIdx.x;
    const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    const float sc = ga * is, sh = be - mu * sc;                 // the forward pass's expression decides the activation's branch
    float4 dz[SMALL_K], xh[SMALL_K];
    int32_t od[SMALL_K];   // element offsets: these tensors are far below 2^31 elements (bn_small_ok)
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < SMALL_K; ++k) {
        const int64_t g = (int64_t)threadIdx.x + (int64_t)k * SMALL_NT;
        od[k] = -1;
        dz[k] = make_float4(0.f, 0.f, 0.f, 0.f); xh[k] = dz[k];
        if (g < m.per_chan) {
            const Pos p = locate_c(m, c, (uint32_t)g);
            const float4 v = *reinterpret_cast<const float4*>(x + offs_c(m, xv, p));
            const float4 d = *reinterpret_cast<const float4*>(dy + offs_c(m, dyv, p));
            od[k] = (int32_t)offs_c(m, dxv, p);
            const float mk = mask ? mask[(int64_t)p.n * m.C + c] : 1.f;
            const float vv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {d.x, d.y, d.z, d.w};
            float zz[4], hh[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hh[i] = (vv[i] - mu) * is;
                const float z = (vv[i] * sc + sh) * mk;
                float t = dd[i] * mk;
                if (act == DCV_ACT_LEAKY) t *= (z > 0.f ? 1.f : slope);
                else if (act == DCV_ACT_TANH) { const float th = tanhf(z); t *= 1.f - th * th; }
                zz[i] = t;
                a0 += t; a1 += t * hh[i];
            }
            dz[k] = make_float4(zz[0], zz[1], zz[2], zz[3]);
            xh[k] = make_float4(hh[0], hh[1], hh[2], hh[3]);
        }
    }
    double acc[2] = {(double)a0, (double)a1};
    block_sum_1024(acc, red);
    if (threadIdx.x == 0) { dbeta[c] = (float)acc[0]; dgamma[c] = (float)acc[1]; }
    const float k0 = training ? (float)(acc[0] / count) : 0.f, k1 = training ? (float)(acc[1] / count) : 0.f;   // bn_bwd_finalize_kernel's coef
#pragma unroll
    for (int k = 0; k < SMALL_K; ++k)
        if (od[k] >= 0) {
            float4 o;
            o.x = ga * is * (dz[k].x - k0 - xh[k].x * k1);
            o.y = ga * is * (dz[k].y - k0 - xh[k].y * k1);
            o.z = ga * is * (dz[k].z - k0 - xh[k].z * k1);
            o.w = ga * is * (dz[k].w - k0 - xh[k].w * k1);
            *reinterpret_cast<float4*>(dx + od[k]) = o;
        }
}

static bool bn_small_ok(const RowMap& m, const ChanMap& cm) {
    static const bool off = getenv("DCV_NO_BN_SMALL") != nullptr;
    return !off && m.vec == 4 && m.inner != 1 && cm.per_chan <= (int64_t)SMALL_NT * SMALL_K && cm.per_chan >= 64 && m.C >= 32 && m.groups < (1ll << 28);
}

// ------------------------------------------------------------------------- //
// GAN loss: one block, value + gradient
// ------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void gan_loss_kernel(const float* __restrict__ y, int64_t n, int kind, float* __restrict__ loss_out, int accumulate, float* __restrict__ dy) {
    __shared__ double red[4];
    const float inv = 1.f / (float)n;
    double acc[1] = {0.0};
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float v = y[i];
        float f, g;
        if (kind == 0 || kind == 4) {        // softplus(-v)
            f = fmaxf(-v, 0.f) + log1pf(expf(-fabsf(v)));
            g = -1.f / (1.f + expf(v));
        } else if (kind == 1) {              // softplus(v)
            f = fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));
            g = 1.f / (1.f + expf(-v));
        } else if (kind == 2) {              // relu(1 - v)
            f = fmaxf(1.f - v, 0.f);
            g = (1.f - v > 0.f) ? -1.f : 0.f;
        } else {                             // relu(1 + v)
            f = fmaxf(1.f + v, 0.f);
            g = (1.f + v > 0.f) ? 1.f : 0.f;
        }
        acc[0] += (double)f;
        if (dy) dy[i] = g * inv;
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) {
        const float val = (float)(acc[0] / (double)n);
        *loss_out = accumulate ? *loss_out + val : val;
    }
}

// ------------------------------------------------------------------------- //
// Adam
// ------------------------------------------------------------------------- //
// torch.optim.Adam's single-tensor update, operation for operation (torch/optim/adam.py _single_tensor_adam):
//   g' = g * gscale + wd * p ; m = lerp(m, g', 1 - b1) ; v = b2 * v + (1 - b2) * g' * g'
//   p -= step_size * m / (sqrt(v) / bc2_sqrt + eps)
// with the scalars (1 - b1, 1 - b2, step_size = lr / (1 - b1^t), bc2_sqrt) computed in double on the host and
// rounded once, as Python does.  fp contraction is off for this file's Adam code path (see adam_update).
struct AdamScalars {
    float w1, b2, w2, eps, wd, step_size, bc2_sqrt, gscale;   // w1 = 1 - beta1, w2 = 1 - beta2
};
__device__ __forceinline__ void adam_update(float& pi, float gi, float& mi, float& vi, const AdamScalars& k) {
#pragma clang fp contract(off)
    gi = gi * k.gscale;
    gi = gi + k.wd * pi;
    // at::lerp: weight < 0.5 ? a + w (b - a) : b - (b - a)(1 - w)
    const float d = gi - mi;
    mi = k.w1 < 0.5f ? mi + k.w1 * d : gi - d * (1.f - k.w1);
    vi = vi * k.b2;
    vi = vi + (k.w2 * gi) * gi;   // addcmul_(g, g, value): value * t1 * t2
    const float denom = sqrtf(vi) / k.bc2_sqrt + k.eps;
    pi = pi + (-k.step_size) * (mi / denom);
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const AdamScalars k) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_update(pi, g[i], mi, vi, k);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}

// up to ADAM_MT tensors of one optimiser in one launch (the DCVGAN models have ~100 small parameter tensors:
// one launch each was ~100 x 5 us of dispatch per step for 70 MB of state)
#define ADAM_MT 24
struct AdamPack {
    float* p[ADAM_MT];
    const float* g[ADAM_MT];
    float* m[ADAM_MT];
    float* v[ADAM_MT];
    int64_t n[ADAM_MT];
    int32_t first_block[ADAM_MT + 1];   // prefix sum of the tensors' block counts (4096 elements per block)
};
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamPack k, int nt, const AdamScalars sc) {
    int t = 0;
    while (t + 1 < nt && (int)blockIdx.x >= k.first_block[t + 1]) ++t;
    const int64_t base = (int64_t)((int)blockIdx.x - k.first_block[t]) * 4096;
    float* __restrict__ p = k.p[t];
    const float* __restrict__ g = k.g[t];
    float* __restrict__ m = k.m[t];
    float* __restrict__ v = k.v[t];
    const int64_t n = k.n[t];
#pragma unroll 4
    for (int e = 0; e < 16; ++e) {
        const int64_t i = base + e * 256 + threadIdx.x;
        if (i >= n) break;
        float pi = p[i], mi = m[i], vi = v[i];
        adam_update(pi, g[i], mi, vi, sc);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}

// ------------------------------------------------------------------------- //
// GRU recurrence (dm <= 32): one 64-lane block per sample, lane u < dm owns unit u
// ------------------------------------------------------------------------- //
#define GRU_MAXD 32
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(64) void gru_fwd_kernel(const float* __restrict__ e, const float* __restrict__ h0, const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                     const float* __restrict__ b_ih, const float* __restrict__ b_hh, float* __restrict__ out, float* __restrict__ gates,
                                                     int T, int B, int dm) {
    __shared__ float h[GRU_MAXD], x[GRU_MAXD];
    const int b = blockIdx.x, u = threadIdx.x;
    if (u < dm) h[u] = h0[(int64_t)b * dm + u];
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (u < dm) x[u] = e[((int64_t)t * B + b) * dm + u];
        __syncthreads();
        float hn_new = 0.f;
        if (u < dm) {
            float ir = b_ih[u], iz = b_ih[dm + u], in_ = b_ih[2 * dm + u];
            float hr = b_hh[u], hz = b_hh[dm + u], hn = b_hh[2 * dm + u];
            for (int k = 0; k < dm; ++k) {
                const float xk = x[k], hk = h[k];
                ir += w_ih[(int64_t)u * dm + k] * xk;
                iz += w_ih[(int64_t)(dm + u) * dm + k] * xk;
                in_ += w_ih[(int64_t)(2 * dm + u) * dm + k] * xk;
                hr += w_hh[(int64_t)u * dm + k] * hk;
                hz += w_hh[(int64_t)(dm + u) * dm + k] * hk;
                hn += w_hh[(int64_t)(2 * dm + u) * dm + k] * hk;
            }
            const float r = sigmoidf_(ir + hr), z = sigmoidf_(iz + hz);
            const float n = tanhf(in_ + r * hn);
            hn_new = n + z * (h[u] - n);
            float* gt = gates + ((int64_t)t * B + b) * 4 * dm;
            gt[u] = r; gt[dm + u] = z; gt[2 * dm + u] = n; gt[3 * dm + u] = hn;
            out[((int64_t)b * T + t) * dm + u] = hn_new;
        }
        __syncthreads();
        if (u < dm) h[u] = hn_new;
        __syncthreads();
    }
}

// per-sample BPTT; partial parameter gradients go to part[b][P], P = 6*dm*dm + 6*dm,
// layout [dw_ih (3dm*dm) | dw_hh (3dm*dm) | db_ih (3dm) | db_hh (3dm)]
__global__ __launch_bounds__(64) void gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ e, const float* __restrict__ h0, const float* __restrict__ out,
                                                     const float* __restrict__ gates, const float* __restrict__ w_hh, float* __restrict__ part, int T, int B, int dm, int lds_acc) {
    __shared__ float dgh[3 * GRU_MAXD], dh[GRU_MAXD];
    extern __shared__ float gru_acc[];   // lds_acc: the sample's P partial sums (the same additions in the same order, just not through global memory)
    const int b = blockIdx.x, u = threadIdx.x;
    const int P = 6 * dm * dm + 6 * dm;
    float* pp = lds_acc ? gru_acc : part + (int64_t)b * P;
    for (int i = u; i < P; i += 64) pp[i] = 0.f;
    if (u < dm) dh[u] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        float dr_pre = 0.f, dz_pre = 0.f, dn_pre = 0.f, dhn = 0.f, dh_keep = 0.f;
        if (u < dm) {
            const float* gt = gates + ((int64_t)t * B + b) * 4 * dm;
            const float r = gt[u], z = gt[dm + u], n = gt[2 * dm + u], hn = gt[3 * dm + u];
            const float hprev = t > 0 ? out[((int64_t)b * T + t - 1) * dm + u] : h0[(int64_t)b * dm + u];
            const float d = dout[((int64_t)b * T + t) * dm + u] + dh[u];
            const float dn = d * (1.f - z);
            const float dz = d * (hprev - n);
            dh_keep = d * z;
            dn_pre = dn * (1.f - n * n);
            dhn = dn_pre * r;
            dr_pre = dn_pre * hn * r * (1.f - r);
            dz_pre = dz * z * (1.f - z);
            dgh[u] = dr_pre; dgh[dm + u] = dz_pre; dgh[2 * dm + u] = dhn;
            // parameter gradients owned by this lane: rows u, dm+u, 2dm+u
            for (int k = 0; k < dm; ++k) {
                const float xk = e[((int64_t)t * B + b) * dm + k];
                const float hk = t > 0 ? out[((int64_t)b * T + t - 1) * dm + k] : h0[(int64_t)b * dm + k];
                pp[(int64_t)u * dm + k] += dr_pre * xk;
                pp[(int64_t)(dm + u) * dm + k] += dz_pre * xk;
                pp[(int64_t)(2 * dm + u) * dm + k] += dn_pre * xk;
                pp[3 * dm * dm + (int64_t)u * dm + k] += dr_pre * hk;
                pp[3 * dm * dm + (int64_t)(dm + u) * dm + k] += dz_pre * hk;
                pp[3 * dm * dm + (int64_t)(2 * dm + u) * dm + k] += dhn * hk;
            }
            float* pb = pp + 6 * dm * dm;
            pb[u] += dr_pre; pb[dm + u] += dz_pre; pb[2 * dm + u] += dn_pre;
            pb[3 * dm + u] += dr_pre; pb[4 * dm + u] += dz_pre; pb[5 * dm + u] += dhn;
        }
        __syncthreads();
        if (u < dm) {
            float s = dh_keep;
            for (int j = 0; j < 3 * dm; ++j) s += w_hh[(int64_t)j * dm + u] * dgh[j];
            dh[u] = s;
        }
        __syncthreads();
    }
    if (lds_acc) {
        float* out_p = part + (int64_t)b * P;
        for (int i = u; i < P; i += 64) out_p[i] = pp[i];
    }
}

__global__ void gru_reduce_kernel(const float* __restrict__ part, int B, int P, int dm, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[(int64_t)b * P + i];
    const int w = 3 * dm * dm;
    if (i < w) dw_ih[i] = s;
    else if (i < 2 * w) dw_hh[i - w] = s;
    else if (i < 2 * w + 3 * dm) db_ih[i - 2 * w] = s;
    else db_hh[i - 2 * w - 3 * dm] = s;
}


// ------------------------------------------------------------------------- //
// sampling path (util.py:58-79, 198-248): float videos -> uint8 on the device
// ------------------------------------------------------------------------- //
// out[b][c*rep + q][t][h][w] = uint8( (clip(x,-1,1) + 1) / 2 * 255 )   (numpy astype: truncation),
// the same fp32 operation order as the reference so the bytes are identical.  rep = 3 tiles a
// 1-channel depth video into RGB (util.py:219-222).
struct ToU8 {
    const float* x; uint8_t* out; RowView xv; int rep;
    template <int VEC> __device__ void apply(const RowMap& m, const Pos& p) const {
        float v[VEC];
        ld<VEC>(x + offs(m, xv, p), v);
        const int64_t plane = (int64_t)m.H * m.W;
        const int64_t col = m.inner == 1 ? (int64_t)p.h * m.W + p.w : p.col;
        for (int q = 0; q < rep; ++q) {
            uint8_t* o = out + ((((int64_t)p.n * m.C + p.c) * rep + q) * m.D + p.d) * plane + col;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float t = fminf(fmaxf(v[i], -1.f), 1.f);
                t = __fmul_rn(__fmul_rn(__fadd_rn(t, 1.f), 0.5f), 255.f);
                o[i] = (uint8_t)(int)t;
            }
        }
    }
};

// per-frame min / max of the flow magnitude (one block per (b, t) frame)
__global__ __launch_bounds__(256) void flow_minmax_kernel(const float* __restrict__ f, RowView fv, int T, int HW, int W, float scale, float* __restrict__ mm) {
    __shared__ float smin[4], smax[4];
    const int b = blockIdx.x / T, t = blockIdx.x % T;
    float lo = 3.4e38f, hi = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const int64_t o = (int64_t)b * fv.sn + (int64_t)t * fv.sd + (int64_t)(i / W) * fv.sh + (int64_t)(i % W) * fv.sw;
        const float x = fminf(fmaxf(f[o], -1.f), 1.f) * scale, y = fminf(fmaxf(f[o + fv.sc], -1.f), 1.f) * scale;
        const float mg = sqrtf(x * x + y * y);
        lo = fminf(lo, mg); hi = fmaxf(hi, mg);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm[2 * blockIdx.x] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        mm[2 * blockIdx.x + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

// util.visualize_optical_flow (util.py:143-170): hue = direction (OpenCV 8-bit H = degrees / 2),
// saturation 255, value = magnitude min-max normalised per frame; HSV -> RGB, uint8 (B,3,T,H,W).
__global__ __launch_bounds__(256) void flow_to_rgb_kernel(const float* __restrict__ f, RowView fv, int B, int T, int HW, int W, float scale,
                                                          const float* __restrict__ mm, uint8_t* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * T * HW) return;
    const int i = (int)(idx % HW);
    const int bt = (int)(idx / HW), b = bt / T, t = bt % T;
    const int64_t o = (int64_t)b * fv.sn + (int64_t)t * fv.sd + (int64_t)(i / W) * fv.sh + (int64_t)(i % W) * fv.sw;
    const float x = fminf(fmaxf(f[o], -1.f), 1.f) * scale, y = fminf(fmaxf(f[o + fv.sc], -1.f), 1.f) * scale;
    const float mg = sqrtf(x * x + y * y);
    float ang = atan2f(y, x);
    if (ang < 0.f) ang += 6.283185307179586f;
    const int Hh = (int)(ang * 180.f / 3.14159265358979f / 2.f) & 255;        // uint8 assignment truncates
    const float lo = mm[2 * bt], hi = mm[2 * bt + 1];
    const int V = hi > lo ? (int)((mg - lo) * (255.f / (hi - lo))) : 0;
    // HSV (H in [0,180), S = 255, V) -> RGB
    const float h6 = (float)Hh / 30.f;
    const int sec = ((int)h6) % 6;
    const float fr = h6 - floorf(h6), v = (float)V;
    const float p_ = 0.f, q_ = v * (1.f - fr), t_ = v * fr;
    float r, g, bl;
    switch (sec) {
        case 0: r = v; g = t_; bl = p_; break;
        case 1: r = q_; g = v; bl = p_; break;
        case 2: r = p_; g = v; bl = t_; break;
        case 3: r = p_; g = q_; bl = v; break;
        case 4: r = t_; g = p_; bl = v; break;
        default: r = v; g = p_; bl = q_; break;
    }
    const int64_t plane = (int64_t)T * HW;
    uint8_t* ob = out + ((int64_t)b * 3) * plane + (int64_t)t * HW + i;
    ob[0] = (uint8_t)(int)(r + 0.5f); ob[plane] = (uint8_t)(int)(g + 0.5f); ob[2 * plane] = (uint8_t)(int)(bl + 0.5f);
}


// ------------------------------------------------------------------------- //
// input pipeline (dataset.py:125-186): frames as read from disk -> training tensors
// ------------------------------------------------------------------------- //
// out[b][c][t][h][w] = float(in[b][t][h][w][c]) / div - sub   (numpy's fp32 operation order)
template <class TIN>
__global__ __launch_bounds__(256) void decode_video_kernel(const TIN* __restrict__ in, float* __restrict__ out, int64_t BT_per_B /*T*/, int64_t HW, int Cc, int64_t total, float div, float sub) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // index over (b, c, t, hw) of the output
    if (i >= total) return;
    const int64_t hw = i % HW;
    int64_t r = i / HW;
    const int64_t t = r % BT_per_B; r /= BT_per_B;
    const int c = (int)(r % Cc);
    const int64_t b = r / Cc;
    const float v = (float)in[((b * BT_per_B + t) * HW + hw) * Cc + c];
    out[i] = v / div - sub;
}

// SURREAL depth (dataset.py:137-156): foreground = depth < 1e10; min-max of the foreground per clip
__global__ __launch_bounds__(256) void surreal_minmax_kernel(const float* __restrict__ d, int64_t per_clip, float* __restrict__ mm) {
    __shared__ float smin[4], smax[4];
    const float* p = d + (int64_t)blockIdx.x * per_clip;
    float lo = 3.4e38f, hi = -3.4e38f;
    for (int64_t i = threadIdx.x; i < per_clip; i += 256) {
        const float v = p[i];
        if (v < 1e10f) { lo = fminf(lo, v); hi = fmaxf(hi, v); }
    }
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm[2 * blockIdx.x] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        mm[2 * blockIdx.x + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}
__global__ __launch_bounds__(256) void surreal_norm_kernel(const float* __restrict__ d, int64_t per_clip, int64_t total, const float* __restrict__ mm, float* __restrict__ out) {
#pragma clang fp contract(off)   // numpy rounds h*1.8 before subtracting; no fma
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / per_clip;
    const float v = d[i], mi = mm[2 * b], ma = mm[2 * b + 1];
    float o = 1.0f;                                        // background
    if (v < 1e10f) {
        float h = v;
        if (ma - mi > 0.f) h = (v - mi) / (ma - mi);
        o = h * 1.8f - 1.0f;                               // [-1.0, 0.8]
    }
    out[i] = o;
}


// ------------------------------------------------------------------------- //
// segmentation branch: reductions over the channel axis, one thread per position.
// Consecutive threads are consecutive w, so every per-channel access is coalesced.
// ------------------------------------------------------------------------- //
struct PosView {   // (n, c, d, h, w) view: element offset of channel 0 at a position, channel stride
    int64_t sn, sc, sd, sh, sw;
    int32_t C, D, H, W;
};
__device__ __forceinline__ int64_t pos_offset(const PosView& v, int64_t p) {
    const int w = (int)(p % v.W); p /= v.W;
    const int h = (int)(p % v.H); p /= v.H;
    const int d = (int)(p % v.D); p /= v.D;
    return p * v.sn + d * v.sd + h * v.sh + w * v.sw;
}
__device__ __forceinline__ int argmax_first(const float* __restrict__ x, int C, int64_t sc) {
    float best = x[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
        const float v = x[c * sc];
        if (v > best) { best = v; bi = c; }   // strict: the first maximum wins, like torch.argmax / np.argmax
    }
    return bi;
}
// y = softmax(x) over channels (generator.py:75-76 nn.Softmax(dim=1))
__global__ __launch_bounds__(256) void softmax_c_fwd_kernel(const float* __restrict__ x, PosView xv, float* __restrict__ y, PosView yv, int64_t P) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float* xp = x + pos_offset(xv, p);
    float* yp = y + pos_offset(yv, p);
    float mx = xp[0];
    for (int c = 1; c < xv.C; ++c) mx = fmaxf(mx, xp[c * xv.sc]);
    float sum = 0.f;
    for (int c = 0; c < xv.C; ++c) {
        const float e = expf(xp[c * xv.sc] - mx);
        yp[c * yv.sc] = e;
        sum += e;
    }
    for (int c = 0; c < xv.C; ++c) yp[c * yv.sc] = yp[c * yv.sc] / sum;
}
// dx = y * (dy - sum_c dy * y)
__global__ __launch_bounds__(256) void softmax_c_bwd_kernel(const float* __restrict__ dy, PosView dv, const float* __restrict__ y, PosView yv,
                                                           float* __restrict__ dx, PosView xv, int64_t P) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float* dp = dy + pos_offset(dv, p);
    const float* yp = y + pos_offset(yv, p);
    float* xp = dx + pos_offset(xv, p);
    float dot = 0.f;
    for (int c = 0; c < dv.C; ++c) dot += dp[c * dv.sc] * yp[c * yv.sc];
    for (int c = 0; c < dv.C; ++c) xp[c * xv.sc] = yp[c * yv.sc] * (dp[c * dv.sc] - dot);
}
// one-hot (or softmax) maps -> {-1, +1} maps (generator.py:378-385)
__global__ __launch_bounds__(256) void segm_onehot_kernel(const float* __restrict__ x, PosView xv, float* __restrict__ y, PosView yv, int64_t P) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int bi = argmax_first(x + pos_offset(xv, p), xv.C, xv.sc);
    float* yp = y + pos_offset(yv, p);
    for (int c = 0; c < xv.C; ++c) yp[c * yv.sc] = c == bi ? 1.f : -1.f;
}
// argmax -> part colour (util.py:236-246); out uint8 (N, 3, D, H, W) contiguous; palette = C x 3 bytes
__global__ __launch_bounds__(256) void segm_color_kernel(const float* __restrict__ x, PosView xv, const uint8_t* __restrict__ palette,
                                                        uint8_t* __restrict__ out, int64_t P) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int bi = argmax_first(x + pos_offset(xv, p), xv.C, xv.sc);
    const int64_t plane = (int64_t)xv.D * xv.H * xv.W;
    const int64_t n = p / plane, r = p % plane;
    for (int k = 0; k < 3; ++k) out[(n * 3 + k) * plane + r] = palette[bi * 3 + k];
}
// dataset.py:176-181: label frames (T, H, W) uint8 -> one-hot float (C, T, H, W), per clip of a (B, T, H, W) batch
__global__ __launch_bounds__(256) void segm_decode_kernel(const uint8_t* __restrict__ labels, float* __restrict__ out, int C, int64_t per_clip, int64_t P) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int64_t b = p / per_clip, r = p % per_clip;
    const int l = labels[p];
    for (int c = 0; c < C; ++c) out[(b * C + c) * per_clip + r] = c == l ? 1.f : 0.f;
}

}  // namespace dcv

using namespace dcv;

extern "C" {

size_t dcv_bn_workspace_bytes(int channels) {
    // partial sums: C * split * 2 doubles (split <= 2048) + coef (2C floats)
    return (size_t)(2048 + channels) * 2 * sizeof(double) * 2 + (size_t)channels * 2 * sizeof(float) + 512;
}

int dcv_act_forward(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, int act, float slope, void* stream) {
    if (!x || !y || !xd || !yd || !same_shape(*xd, *yd)) return fail(DCV_EINVAL, "act_forward: bad arguments");
    const dcv_dims5* views[2] = {xd, yd};
    const void* ptrs[2] = {x, y};
    RowMap m = make_rowmap(*xd, views, 2, ptrs);
    ActFwd f{x, y, rv(*xd), rv(*yd), act, slope};
    return launch_ew(m, f, static_cast<hipStream_t>(stream));
}

int dcv_act_backward(const float* dy, const dcv_dims5* dyd, const float* y, const dcv_dims5* yd, float* dx, const dcv_dims5* dxd, int act, float slope, void* stream) {
    if (!dy || !y || !dx || !same_shape(*dyd, *yd) || !same_shape(*dyd, *dxd)) return fail(DCV_EINVAL, "act_backward: bad arguments");
    const dcv_dims5* views[3] = {dyd, yd, dxd};
    const void* ptrs[3] = {dy, y, dx};
    RowMap m = make_rowmap(*dyd, views, 3, ptrs);
    ActBwd f{dy, y, dx, rv(*dyd), rv(*yd), rv(*dxd), act, slope};
    return launch_ew(m, f, static_cast<hipStream_t>(stream));
}

int dcv_axpby(const float* x, const dcv_dims5* xd, float a, const float* z, const dcv_dims5* zd, float b, float* y, const dcv_dims5* yd, void* stream) {
    if (!x || !y || !xd || !yd || !same_shape(*xd, *yd) || (z && (!zd || !same_shape(*xd, *zd)))) return fail(DCV_EINVAL, "axpby: bad arguments");
    const dcv_dims5* views[3] = {xd, yd, z ? zd : xd};
    const void* ptrs[3] = {x, y, z ? z : x};
    RowMap m = make_rowmap(*xd, views, 3, ptrs);
    Axpby f{x, z, y, rv(*xd), z ? rv(*zd) : rv(*xd), rv(*yd), a, b};
    return launch_ew(m, f, static_cast<hipStream_t>(stream));
}

int dcv_noise_add(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, float sigma, uint64_t seed, uint64_t offset, void* stream) {
    if (!x || !y || !same_shape(*xd, *yd)) return fail(DCV_EINVAL, "noise_add: bad arguments");
    const dcv_dims5* views[2] = {xd, yd};
    const void* ptrs[2] = {x, y};
    RowMap m = make_rowmap(*xd, views, 2, ptrs);
    NoiseAdd f{x, y, rv(*xd), rv(*yd), sigma, seed, offset};
    return launch_ew(m, f, static_cast<hipStream_t>(stream));
}

int dcv_normal_fill(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    if (!out || n < 0) return fail(DCV_EINVAL, "normal_fill: bad arguments");
    if (n == 0) return DCV_OK;
    const int64_t q = (n + 3) / 4;
    hipLaunchKernelGGL(normal_fill_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), out, n, seed, offset);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_normal_fill_many(float* out, int64_t n, int64_t count, uint64_t seed, uint64_t offset, void* stream) {
    if (!out || n < 0 || count < 0) return fail(DCV_EINVAL, "normal_fill_many: bad arguments");
    if (n == 0 || count == 0) return DCV_OK;
    const int64_t t = (n + 3) / 4 * count;
    if ((t + 255) / 256 >= (1ll << 31)) return fail(DCV_EUNSUPPORTED, "normal_fill_many: too many values for one launch");
    hipLaunchKernelGGL(normal_fill_many_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), out, n, count, seed, offset);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
    if (!mask || n < 0 || p < 0.f || p >= 1.f) return fail(DCV_EINVAL, "dropout_mask: bad arguments");
    if (n == 0) return DCV_OK;
    const int64_t q = (n + 3) / 4;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), mask, n, p, seed, offset);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_bn_act_forward(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd, const float* mask,
                       int training, float momentum, float eps, int act, float slope, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !y || !gamma || !beta || !save_mean || !save_invstd || !same_shape(*xd, *yd)) return fail(DCV_EINVAL, "bn_act_forward: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int C = xd->c;
    const dcv_dims5* views[2] = {xd, yd};
    const void* ptrs[2] = {x, y};
    RowMap m = make_rowmap(*xd, views, 2, ptrs);
    if (m.groups >= (1ll << 32)) return fail(DCV_EUNSUPPORTED, "bn: tensor too large");
    if (training) {
        if (ws_bytes < dcv_bn_workspace_bytes(C) || !ws) return fail(DCV_EWORKSPACE, "bn_act_forward: workspace too small");
        ChanMap cm = make_chanmap(m);
        if (bn_small_ok(m, cm)) {
            const double cnt = (double)xd->n * xd->d * xd->h * xd->w;
            hipLaunchKernelGGL(bn_fwd_small_kernel, dim3(C), dim3(SMALL_NT), 0, s, cm, x, rv(*xd), y, rv(*yd), gamma, beta, running_mean, running_var, num_batches_tracked,
                               save_mean, save_invstd, mask, cnt, eps, momentum, act, slope);
            DCV_LAUNCH_CHECK();
            return DCV_OK;
        }
        double* partial = static_cast<double*>(ws);
        if (m.vec == 4) hipLaunchKernelGGL((bn_stats_kernel<4>), dim3(C * cm.split), dim3(256), 0, s, cm, x, rv(*xd), partial);
        else hipLaunchKernelGGL((bn_stats_kernel<1>), dim3(C * cm.split), dim3(256), 0, s, cm, x, rv(*xd), partial);
        DCV_LAUNCH_CHECK();
        const double count = (double)xd->n * xd->d * xd->h * xd->w;
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partial, C, cm.split, count, eps, momentum, save_mean, save_invstd, running_mean, running_var, num_batches_tracked);
        DCV_LAUNCH_CHECK();
    } else {
        if (!running_mean || !running_var) return fail(DCV_EINVAL, "bn_act_forward: eval mode needs running stats");
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, s, C, eps, running_mean, running_var, save_mean, save_invstd);
        DCV_LAUNCH_CHECK();
    }
    BnApply f{x, y, rv(*xd), rv(*yd), gamma, beta, save_mean, save_invstd, mask, act, slope};
    return launch_ew(m, f, s);
}

int dcv_bn_act_forward_stats(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd, const float* mask,
                             float momentum, float eps, int act, float slope, const float* stat, int nparts, int pitch,
                             void* ws, size_t ws_bytes, void* stream) {
    if (!x || !y || !gamma || !beta || !save_mean || !save_invstd || !stat || nparts < 1 || !same_shape(*xd, *yd) || pitch < xd->c)
        return fail(DCV_EINVAL, "bn_act_forward_stats: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int C = xd->c;
    const dcv_dims5* views[2] = {xd, yd};
    const void* ptrs[2] = {x, y};
    RowMap m = make_rowmap(*xd, views, 2, ptrs);
    if (m.groups >= (1ll << 32)) return fail(DCV_EUNSUPPORTED, "bn: tensor too large");
    (void)ws; (void)ws_bytes;
    const double count = (double)xd->n * xd->d * xd->h * xd->w;
    {
        const ChanMap cm = make_chanmap(m);
        if (bn_small_ok(m, cm)) {   // the channel fits one workgroup's registers: its own sums are as cheap as combining the conv's partials, and it is one launch
            hipLaunchKernelGGL(bn_fwd_small_kernel, dim3(C), dim3(SMALL_NT), 0, s, cm, x, rv(*xd), y, rv(*yd), gamma, beta, running_mean, running_var, num_batches_tracked,
                               save_mean, save_invstd, mask, count, eps, momentum, act, slope);
            DCV_LAUNCH_CHECK();
            return DCV_OK;
        }
    }
    hipLaunchKernelGGL(bn_partials_finalize_kernel, dim3(C), dim3(256), 0, s, stat, nparts, pitch, count, eps, momentum, save_mean, save_invstd, running_mean, running_var,
                       num_batches_tracked);
    DCV_LAUNCH_CHECK();
    BnApply f{x, y, rv(*xd), rv(*yd), gamma, beta, save_mean, save_invstd, mask, act, slope};
    return launch_ew(m, f, s);
}

// The two halves of dcv_bn_act_forward_stats as calls of their own, for a BatchNorm group whose OUTPUT is never written because its consumers normalise on load
// (dcv_conv_forward_bn / dcv_conv_backward_weight_bn / dcv_conv_backward_data_bn): statistics + running-statistics update only ...
int dcv_bn_forward_stats_only(const float* x, const dcv_dims5* xd, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd,
                              float momentum, float eps, const float* stat, int nparts, int pitch, void* stream) {
    if (!x || !xd || !save_mean || !save_invstd || !stat || nparts < 1 || pitch < xd->c) return fail(DCV_EINVAL, "bn_forward_stats_only: bad arguments");
    const double count = (double)xd->n * xd->d * xd->h * xd->w;
    hipLaunchKernelGGL(bn_partials_finalize_kernel, dim3(xd->c), dim3(256), 0, static_cast<hipStream_t>(stream), stat, nparts, pitch, count, eps, momentum, save_mean, save_invstd,
                       running_mean, running_var, num_batches_tracked);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

// ... and the apply pass alone (y = act(mask * (x * sc + sh)) from saved statistics): the fallback that materialises such an output after all
int dcv_bn_apply(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                 const float* mask, int act, float slope, void* stream) {
    if (!x || !y || !gamma || !beta || !save_mean || !save_invstd || !same_shape(*xd, *yd)) return fail(DCV_EINVAL, "bn_apply: bad arguments");
    const dcv_dims5* views[2] = {xd, yd};
    const void* ptrs[2] = {x, y};
    RowMap m = make_rowmap(*xd, views, 2, ptrs);
    if (m.groups >= (1ll << 32)) return fail(DCV_EUNSUPPORTED, "bn: tensor too large");
    BnApply f{x, y, rv(*xd), rv(*yd), gamma, beta, save_mean, save_invstd, mask, act, slope};
    return launch_ew(m, f, static_cast<hipStream_t>(stream));
}

int dcv_bn_act_backward(const float* dy, const dcv_dims5* dyd, const float* x, const dcv_dims5* xd, float* dx, const dcv_dims5* dxd,
                        const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, const float* mask,
                        int training, int act, float slope, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !dx || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta || !same_shape(*dyd, *xd) || !same_shape(*dyd, *dxd))
        return fail(DCV_EINVAL, "bn_act_backward: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int C = xd->c;
    if (ws_bytes < dcv_bn_workspace_bytes(C) || !ws) return fail(DCV_EWORKSPACE, "bn_act_backward: workspace too small");
    const dcv_dims5* views[3] = {dyd, xd, dxd};
    const void* ptrs[3] = {dy, x, dx};
    RowMap m = make_rowmap(*xd, views, 3, ptrs);
    if (m.groups >= (1ll << 32)) return fail(DCV_EUNSUPPORTED, "bn: tensor too large");
    ChanMap cm = make_chanmap(m);
    if (bn_small_ok(m, cm)) {
        const double cnt = (double)xd->n * xd->d * xd->h * xd->w;
        hipLaunchKernelGGL(bn_bwd_small_kernel, dim3(C), dim3(SMALL_NT), 0, s, cm, dy, rv(*dyd), x, rv(*xd), dx, rv(*dxd), gamma, beta, save_mean, save_invstd, mask,
                           cnt, act, slope, training, dgamma, dbeta);
        DCV_LAUNCH_CHECK();
        return DCV_OK;
    }
    double* partial = static_cast<double*>(ws);
    float* coef = reinterpret_cast<float*>(static_cast<char*>(ws) + (size_t)(2048 + C) * 2 * sizeof(double) * 2);
    if (m.vec == 4 && m.inner != 1 && m.gpr % 256 == 0 && !ew_rows_off())
        hipLaunchKernelGGL(bn_bwd_reduce_rows_kernel, dim3(C * cm.split), dim3(256), 0, s, cm, dy, rv(*dyd), x, rv(*xd), gamma, beta, save_mean, save_invstd, mask, act, slope, partial);
    else if (m.vec == 4)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<4>), dim3(C * cm.split), dim3(256), 0, s, cm, dy, rv(*dyd), x, rv(*xd), gamma, beta, save_mean, save_invstd, mask, act, slope, partial);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<1>), dim3(C * cm.split), dim3(256), 0, s, cm, dy, rv(*dyd), x, rv(*xd), gamma, beta, save_mean, save_invstd, mask, act, slope, partial);
    DCV_LAUNCH_CHECK();
    const double count = (double)xd->n * xd->d * xd->h * xd->w;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partial, C, cm.split, count, dgamma, dbeta, coef);
    DCV_LAUNCH_CHECK();
    BnBwdApply f{dy, x, dx, rv(*dyd), rv(*xd), rv(*dxd), gamma, beta, save_mean, save_invstd, mask, coef, act, slope, training};
    return launch_ew(m, f, s);
}

// y[i] = x[i] * *s: the chain rule through a loss term — the upstream cotangent is a 0-d DEVICE tensor (loss.backward(), trainer.py:319,356)
__global__ __launch_bounds__(256) void scale_dev_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ s, float* __restrict__ y) {
    const float k = *s;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * k;
}

int dcv_scale_dev(const float* x, int64_t n, const float* s, float* y, void* stream) {
    if (!x || !s || !y || n < 0) return fail(DCV_EINVAL, "scale_dev: bad arguments");
    if (n == 0) return DCV_OK;
    hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, s, y);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_gan_loss(const float* y, int64_t n, int kind, float* loss_out, int accumulate, float* dy_out, void* stream) {
    if (!y || !loss_out || n < 1 || kind < 0 || kind > 4) return fail(DCV_EINVAL, "gan_loss: bad arguments");
    hipLaunchKernelGGL(gan_loss_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), y, n, kind, loss_out, accumulate, dy_out);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

// bias corrections and step size in double, rounded once (torch computes them as Python floats)
static AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale) {
    AdamScalars k;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    k.w1 = (float)(1.0 - beta1);
    k.b2 = (float)beta2;
    k.w2 = (float)(1.0 - beta2);
    k.eps = (float)eps;
    k.wd = (float)weight_decay;
    k.step_size = (float)(lr / bc1);
    k.bc2_sqrt = (float)sqrt(bc2);
    k.gscale = (float)grad_scale;
    return k;
}

int dcv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                  int step, double grad_scale, void* stream) {
    if (!p || !g || !m || !v || n < 0 || step < 1) return fail(DCV_EINVAL, "adam_step: bad arguments");
    if (n == 0) return DCV_OK;
    const AdamScalars k = adam_scalars(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n, k);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}


int dcv_adam_step_multi(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* numel,
                        double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!p || !g || !m || !v || !numel)) || step < 1) return fail(DCV_EINVAL, "adam_step_multi: bad arguments");
    const AdamScalars sc = adam_scalars(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    for (int t0 = 0; t0 < n_tensors; t0 += ADAM_MT) {
        AdamPack k;
        memset(&k, 0, sizeof(k));
        const int nt = std::min(ADAM_MT, n_tensors - t0);
        int blocks = 0;
        for (int t = 0; t < nt; ++t) {
            if (!p[t0 + t] || !g[t0 + t] || !m[t0 + t] || !v[t0 + t] || numel[t0 + t] < 0) return fail(DCV_EINVAL, "adam_step_multi: null tensor");
            k.p[t] = p[t0 + t]; k.g[t] = g[t0 + t]; k.m[t] = m[t0 + t]; k.v[t] = v[t0 + t]; k.n[t] = numel[t0 + t];
            k.first_block[t] = blocks;
            blocks += (int)((numel[t0 + t] + 4095) / 4096);
        }
        k.first_block[nt] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), k, nt, sc);
        DCV_LAUNCH_CHECK();
    }
    return DCV_OK;
}

int dcv_decode_video(const void* in, int in_is_u8, int B, int T, int H, int W, int Cc, float div, float sub, float* out, void* stream) {
    if (!in || !out || B < 1 || T < 1 || H < 1 || W < 1 || Cc < 1 || div == 0.f) return fail(DCV_EINVAL, "decode_video: bad arguments");
    const int64_t total = (int64_t)B * Cc * T * H * W;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (in_is_u8) hipLaunchKernelGGL((decode_video_kernel<uint8_t>), dim3(grid), dim3(256), 0, s, static_cast<const uint8_t*>(in), out, (int64_t)T, (int64_t)H * W, Cc, total, div, sub);
    else hipLaunchKernelGGL((decode_video_kernel<float>), dim3(grid), dim3(256), 0, s, static_cast<const float*>(in), out, (int64_t)T, (int64_t)H * W, Cc, total, div, sub);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_surreal_depth(const float* depth, int B, int T, int H, int W, float* out, float* ws_minmax, void* stream) {
    if (!depth || !out || !ws_minmax || B < 1) return fail(DCV_EINVAL, "surreal_depth: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t per = (int64_t)T * H * W, total = per * B;
    hipLaunchKernelGGL(surreal_minmax_kernel, dim3(B), dim3(256), 0, s, depth, per, ws_minmax);
    DCV_LAUNCH_CHECK();
    hipLaunchKernelGGL(surreal_norm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, depth, per, total, ws_minmax, out);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

static PosView posview(const dcv_dims5& d) { return PosView{d.sn, d.sc, d.sd, d.sh, d.sw, d.c, d.d, d.h, d.w}; }
static bool same_positions(const dcv_dims5& a, const dcv_dims5& b) { return a.n == b.n && a.d == b.d && a.h == b.h && a.w == b.w; }

int dcv_softmax_channels_forward(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, void* stream) {
    if (!x || !y || !xd || !yd || !same_shape(*xd, *yd) || xd->c < 1) return fail(DCV_EINVAL, "softmax_channels_forward: bad arguments");
    const int64_t P = (int64_t)xd->n * xd->d * xd->h * xd->w;
    hipLaunchKernelGGL(softmax_c_fwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, posview(*xd), y, posview(*yd), P);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_softmax_channels_backward(const float* dy, const dcv_dims5* dyd, const float* y, const dcv_dims5* yd, float* dx, const dcv_dims5* dxd, void* stream) {
    if (!dy || !y || !dx || !dyd || !yd || !dxd || !same_shape(*dyd, *yd) || !same_shape(*dyd, *dxd)) return fail(DCV_EINVAL, "softmax_channels_backward: bad arguments");
    const int64_t P = (int64_t)yd->n * yd->d * yd->h * yd->w;
    hipLaunchKernelGGL(softmax_c_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, posview(*dyd), y, posview(*yd), dx, posview(*dxd), P);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_segm_onehot(const float* x, const dcv_dims5* xd, float* y, const dcv_dims5* yd, void* stream) {
    if (!x || !y || !xd || !yd || !same_shape(*xd, *yd) || xd->c < 1) return fail(DCV_EINVAL, "segm_onehot: bad arguments");
    const int64_t P = (int64_t)xd->n * xd->d * xd->h * xd->w;
    hipLaunchKernelGGL(segm_onehot_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, posview(*xd), y, posview(*yd), P);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_segm_to_rgb(const float* x, const dcv_dims5* xd, const uint8_t* palette, uint8_t* out, void* stream) {
    if (!x || !xd || !palette || !out || xd->c < 1) return fail(DCV_EINVAL, "segm_to_rgb: bad arguments");
    const int64_t P = (int64_t)xd->n * xd->d * xd->h * xd->w;
    hipLaunchKernelGGL(segm_color_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, posview(*xd), palette, out, P);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_decode_segmentation(const uint8_t* labels, int B, int T, int H, int W, int C, float* out, void* stream) {
    if (!labels || !out || B < 1 || T < 1 || H < 1 || W < 1 || C < 1 || C > 256) return fail(DCV_EINVAL, "decode_segmentation: bad arguments");
    const int64_t per = (int64_t)T * H * W, P = per * B;
    hipLaunchKernelGGL(segm_decode_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), labels, out, C, per, P);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_videos_to_uint8(const float* x, const dcv_dims5* xd, uint8_t* out, int channel_repeat, void* stream) {
    if (!x || !xd || !out || channel_repeat < 1 || channel_repeat > 4) return fail(DCV_EINVAL, "videos_to_uint8: bad arguments");
    const dcv_dims5* views[1] = {xd};
    const void* ptrs[1] = {x};
    RowMap m = make_rowmap(*xd, views, 1, ptrs);
    ToU8 f{x, out, rv(*xd), channel_repeat};
    return launch_ew(m, f, static_cast<hipStream_t>(stream));
}

int dcv_flow_to_rgb(const float* flow, const dcv_dims5* fd, float scale, uint8_t* out, float* ws_minmax, void* stream) {
    if (!flow || !fd || !out || !ws_minmax || fd->c != 2) return fail(DCV_EINVAL, "flow_to_rgb: needs a (B,2,T,H,W) flow video");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = fd->n, T = fd->d, HW = fd->h * fd->w;
    hipLaunchKernelGGL(flow_minmax_kernel, dim3(B * T), dim3(256), 0, s, flow, rv(*fd), T, HW, fd->w, scale, ws_minmax);
    DCV_LAUNCH_CHECK();
    const int64_t tot = (int64_t)B * T * HW;
    hipLaunchKernelGGL(flow_to_rgb_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, flow, rv(*fd), B, T, HW, fd->w, scale, ws_minmax, out);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

size_t dcv_gru_workspace_bytes(int B, int dm) { return (size_t)B * (6 * dm * dm + 6 * dm) * sizeof(float) + 256; }

int dcv_gru_forward(const float* e, const float* h0, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                    float* out, float* gates, int T, int B, int dm, void* stream) {
    if (!e || !h0 || !w_ih || !w_hh || !b_ih || !b_hh || !out || !gates || T < 1 || B < 1 || dm < 1 || dm > GRU_MAXD) return fail(DCV_EINVAL, "gru_forward: bad arguments");
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), e, h0, w_ih, w_hh, b_ih, b_hh, out, gates, T, B, dm);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_gru_backward(const float* dout, const float* e, const float* h0, const float* out, const float* gates, const float* w_ih, const float* w_hh,
                     float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int T, int B, int dm, void* ws, size_t ws_bytes, void* stream) {
    (void)w_ih;
    if (!dout || !e || !h0 || !out || !gates || !w_hh || !dw_ih || !dw_hh || !db_ih || !db_hh || dm > GRU_MAXD) return fail(DCV_EINVAL, "gru_backward: bad arguments");
    if (!ws || ws_bytes < dcv_gru_workspace_bytes(B, dm)) return fail(DCV_EWORKSPACE, "gru_backward: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    const int P = 6 * dm * dm + 6 * dm;
    const int lds_acc = (size_t)P * sizeof(float) <= 48 * 1024;   // dm <= 44: the partial sums live in LDS
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(B), dim3(64), lds_acc ? (size_t)P * sizeof(float) : 0, s, dout, e, h0, out, gates, w_hh, part, T, B, dm, lds_acc);
    DCV_LAUNCH_CHECK();
    hipLaunchKernelGGL(gru_reduce_kernel, dim3((P + 255) / 256), dim3(256), 0, s, part, B, P, dm, dw_ih, dw_hh, db_ih, db_hh);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

}  // extern "C"
