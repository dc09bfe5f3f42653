// Implicit-GEMM convolutions for gfx950 on v_mfma_f32_32x32x2_f32.
//
// Every conv / transposed conv / data-gradient of the DCVGAN step is ONE of two
// GEMM shapes over NCDHW tensors (never re-laid-out, strides are honoured):
//
//  gather GEMM   Y[oc, m] = sum_k  Wp[k, oc] * Xg[k, m]       (fprop, dgrad, convT)
//      m  = (n, od, oh, ow) output position (one stride-parity class per launch
//           for the "scatter" forms), k = (reduction channel, tap);
//      Xg = the gathered input, zero outside the tensor;
//      Wp = weights re-packed K-major once per call by pack_weights_kernel.
//      MFMA A operand = weights (rows = oc), B operand = activations (cols = m),
//      so every accumulator register holds 32 consecutive m of one oc row and
//      the NCDHW store is contiguous along w.
//
//  wgrad GEMM    R[dc, j] = sum_m  D[dc, m] * G[j, m]          (weight gradients)
//      dc = channel of the densely indexed tensor, j = (gathered channel, tap),
//      reduction over all positions m, split over blockIdx.y into slabs that a
//      second kernel sums in a fixed order (bitwise reproducible, no atomics).
//
// fp32 MFMA runs at the vector rate (157 TFLOP/s peak, 1/16 of bf16), so the
// kernel is matrix-pipe bound with a large VALU / LDS / HBM margin: staging is
// plain global -> VGPR -> LDS with one barrier per 16-deep K step and tables
// (KEntry) carry all index arithmetic so one kernel serves every geometry.
#include "dcv_common.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

namespace dcv {

thread_local char g_err[512] = {0};
std::atomic<uint64_t> g_launches{0};
thread_local char g_last_kernel[160] = {0};   // diagnostics: the GEMM kernel instance of the calling thread's last conv call
#define DCV_NOTE_KERNEL(...) snprintf(g_last_kernel, sizeof(g_last_kernel), __VA_ARGS__)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Precision of the MFMA products in the LDS-DMA GEMM kernels: 0 = fp32 (v_mfma_f32_32x32x2_f32, the default and the only
// mode the parity tests and the headline benchmark use), 1 = bf16 products with fp32 accumulation
// (v_mfma_f32_32x32x16_bf16): tensors, weights, statistics and accumulators stay fp32 in HBM and LDS, the MFMA fragments
// are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) as they are read from LDS.  Throughput-only mode (BASELINE configs[2], [4]).
std::atomic<int> g_precision{0};
// per-call override from dcv_conv_geom.mfma (1 = fp32, 2 = bf16; 0 = the process default above), valid while a dcv_conv_* entry point runs on this thread
static thread_local int t_precision = -1;
// dcv_conv_backward_weight_acc: the slab reduce of the calling thread's weight-gradient call ADDS its sum to dw (set around the call, read at the three reduce launches)
static thread_local int t_wgrad_acc = 0;
// dcv_conv_backward_data_bn's request to the data-gradient path: when the op is the colour generator's RGB head (widen_mfma_kernel's geometry), run_gather launches
// head_bn_kernel<mode> instead and sets `used`
struct HeadBnRequest {
    int mode, used, nwg;
    const float* bx; dcv_dims5 bxd; float* bdx; dcv_dims5 bdxd;
    const float* gamma; const float* beta; const float* mean; const float* invstd;
    const float* cst; float* partial; size_t partial_bytes;
    int cbn; float slope;
};
static thread_local HeadBnRequest* t_headbn = nullptr;
// "The first cbn channels of the operand are act(BatchNorm(bx)) and were never written": dcv_conv_forward_bn / dcv_conv_backward_weight_bn ask the RGB head's forward
// (thin_rows_kernel) and weight gradient (thinj_wgrad_kernel) to read the BatchNorm INPUT for those channels and normalise + activate on the fly.
struct BnView {
    const float* bx; const float* gamma; const float* beta; const float* mean; const float* invstd;
    int64_t bx_sn;
    int32_t bx_sc, bx_sh, cbn, act;
    float slope; int32_t used;
};
static thread_local BnView* t_bnview = nullptr;
static inline int eff_precision() { return t_precision >= 0 ? t_precision : g_precision.load(std::memory_order_relaxed); }
struct PrecisionScope {
    int saved;
    explicit PrecisionScope(const dcv_conv_geom* g) : saved(t_precision) { if (g && g->mfma > 0) t_precision = g->mfma - 1; }
    ~PrecisionScope() { t_precision = saved; }
};

// four v_cvt_pk_bf16_f32 (pairs converted as 2-vectors and laid side by side as dwords; element-wise conversion made the compiler
// convert some values singly and merge them with v_perm / v_alignbit: 80 conversions + 64 merges per 64-position tile of the
// weight-gradient kernel instead of 64 + 0)
__device__ __forceinline__ bf16x8 pack_bf16x8(const float (&t)[8]) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    u32x4_ w;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2_ f = {t[2 * q], t[2 * q + 1]};
        w[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_));
    }
    return __builtin_bit_cast(bf16x8, w);
}

// fp32 on the bf16 matrix pipe (precision mode 2, "f32x6"): x = hi + mid + lo EXACTLY, each piece the bf16 RNE of what the pieces before it left
// (|mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|; 8 + 8 + 8 significand bits cover fp32's 24).  A bf16 x bf16 product is exact in fp32, so
// sum_k a_k b_k = sum_k sum_{p,q} a_k^(p) b_k^(q); the six terms with p + q <= 2 are formed by v_mfma_f32_32x32x16_bf16 and accumulated in fp32,
// the three dropped ones (mid*lo, lo*mid, lo*lo) are <= 2^-26 of a product each — below fp32's own rounding of it (2^-24).
// 11 VALU operations per pair of elements: v_cvt_pk_bf16_f32, two re-widenings (shift / mask), two subtractions — twice — and the last conversion.
__device__ __forceinline__ void split3_bf16x8(const float (&t)[8], bf16x8 (&o)[3]) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    u32x4_ h, m, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2_ f = {t[2 * q], t[2 * q + 1]};
        const uint32_t hw = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_));
        const f32x2_ r = {f[0] - __builtin_bit_cast(float, hw << 16), f[1] - __builtin_bit_cast(float, hw & 0xffff0000u)};
        const uint32_t mw = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_));
        const f32x2_ r2 = {r[0] - __builtin_bit_cast(float, mw << 16), r[1] - __builtin_bit_cast(float, mw & 0xffff0000u)};
        h[q] = hw; m[q] = mw; l[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2_));
    }
    o[0] = __builtin_bit_cast(bf16x8, h); o[1] = __builtin_bit_cast(bf16x8, m); o[2] = __builtin_bit_cast(bf16x8, l);
}
// The bf16 MFMA's accumulate is not round-to-nearest: what it drops of a sum is dropped toward minus infinity (measured: the mean error of an f32x6 output sits ~1.4 ulp below
// zero at K = 8192 where the fp32 MFMA's is zero; profiles/r04_f32x6_bias.txt), and sums over outputs — BatchNorm statistics, BatchNorm-parameter gradients — collect that.
// Cancelled by alternating the SIGN of what is being accumulated: K steps come in phases of 8; in the odd phases the packed weights are negated (pack_weights_kernel) and the
// accumulators hold MINUS the running sum (negated at each phase change: 16 TOC TM sign flips per 8 steps of 6 TOC TM MFMAs), so the truncation pulls the sum up as often as down.
#ifdef DCV_X6_NO_PHASES      // A/B builds only (tools/ab_lib.sh): the emulation as first built, with its one-sided error
#define X6_PHASE(IT) false
#else
#define X6_PHASE(IT) ((((IT) >> 3) & 1) != 0)
#endif
// the six products, smallest terms first (a = the A operand's pieces, b = the B operand's)
#define DCV_MFMA_X6(ACC, A3, B3)                                                              \
    {                                                                                         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A3)[2], (B3)[0], ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A3)[0], (B3)[2], ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A3)[1], (B3)[1], ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A3)[1], (B3)[0], ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A3)[0], (B3)[1], ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A3)[0], (B3)[0], ACC, 0, 0, 0);        \
    }

#ifdef DCV_STAMP
// diagnostic build only: per-wave cycle totals of the K-loop segments (never in the shipped library)
__device__ unsigned long long g_stamp[4096][4][6];
#define STAMP(var) unsigned long long var = clock64(); __builtin_amdgcn_sched_barrier(0)
#else
#define STAMP(var) __builtin_amdgcn_sched_barrier(0)
#endif

// one row of the K (gather) or J (wgrad) index table
struct KEntry {
    int32_t x_off;    // element offset added to the thread's base: chan*sc + tap deltas
    uint32_t tapsel;  // per-dim tap bits: 1<<ud | 1<<(8+uh) | 1<<(16+uw); bit 31 = padding row
    int32_t w_off;    // element offset of this (channel, tap) inside the weight tensor
    int32_t pad;
};

struct DimTaps {
    int32_t n;          // taps along this dim (<= 8)
    int32_t mul, base;  // gathered position = o * mul + base + delta[u]
    int32_t size;       // extent of the gathered tensor along this dim
    int32_t delta[8];
};

struct GatherArgs {
    const float* x;
    float* y;
    const float* wp;
    const int32_t* koff;    // per K row: BYTE offset added to the thread's base (SoA copy of KEntry.x_off * 4)
    const uint32_t* ksel;   // per K row: tap-selection bits (KEntry.tapsel)
    int32_t M, OC, OCp, KIT;     // positions, out channels, packed pitch, 16-row K iterations
    FastDiv div_sp, div_hw, div_w;  // m -> (n, od, oh, ow): by OD*OH*OW, OH*OW, OW
    int32_t OD, OH, OW, pad0;
    DimTaps td, th, tw;
    int64_t x_sn;
    int32_t x_sd, x_sh, x_sw, x_back;   // x_back: DSTEP base shift, elements
    int64_t y_sn, y_sc, y_sd, y_sh, y_sw, y_off;
    int32_t act, accumulate;
    float slope, pad2;
    // split-K: blockIdx.y handles K steps [y*kper, (y+1)*kper) and writes raw partial sums to
    // slab[y][oc][m] (pitch Mp); splitk_reduce_kernel sums them in order and does the epilogue.
    float* slab;
    int32_t kper, Mp;
    // structured K walk (regular geometries): the 16 rows of a K step have iteration-invariant byte
    // offsets s_local[r] / tap bits s_sel[r]; step `it` adds the scalar (it >> s_log2p) * s_stepA +
    // (it & ((1 << s_log2p) - 1)) * s_stepD.  No index-table reads, no per-element VALU in the loop.
    int32_t structured, s_log2p, s_stepA, s_stepD;
    int32_t s_local[16];
    uint32_t s_sel[16];
    // patch staging (w-contiguous operand, OW | tile): instead of one gathered row per (channel, tap), the raw
    // input rows each output row of the tile needs are copied once — NH rows of IW + 8 words (a zero granule
    // on either side) per (channel, output row) — as 16-byte LDS-DMA granules; the MFMA fragment reads do the
    // gathering (per-lane patch address + per-k-row scalar offset s_local[r]).  2-4 DMAs per wave and step
    // instead of 8-16.
    int32_t patch, p_G, p_log2nh, p_log2tr;   // granules per step; rows per (channel, output row); output rows per tile
    int32_t p_log2ow, p_iwp, p_ihmin, p_iwmin4;
    int32_t p_dw1, p_sc4, slab_mp, p_pad2;    // word step from an even k row to the next (tap uw -> uw + 1); channel stride in bytes; slab_mp: see rag_m0
    FastDiv p_gpr;                            // granules per patch row = IW / 4 + 2
    // BatchNorm statistics of the output, fused into the epilogue: stat[(class * stat_ntm + m tile)][OCp][2] =
    // {sum, sum of squares} over the tile's positions (fp32, <= 256 terms each; combined in fp64 by the BN op)
    float* stat;
    int32_t stat_ntm, stat_cls;
    // data gradients only: y *= act'(gate[same element]) — the (Leaky)ReLU derivative of the layer that produced this
    // conv's input, taken from that input itself (gate has y's shape and strides); nullptr = off
    const float* gate;
    float gate_slope;
    // ragged split-K (LDS-DMA kernel): only the position tiles from rag_m0 on — the ones that would otherwise run as an under-filled
    // last round of the chip's 1024 resident workgroups — are split over blockIdx.y and go through slab[y][oc][m - rag_m0] (pitch
    // slab_mp) + splitk_reduce_kernel; the tiles before it take the direct epilogue and exist for blockIdx.y == 0 only.
    // Plain split-K: rag_m0 = 0, slab_mp = Mp.
    int32_t rag_m0;
};

// up to 4 stride-parity classes of one scatter-form op run as ONE launch (blockIdx.z = class):
// 4x the workgroups per launch fill the chip and amortise the tail of each class.
struct GatherArgsPack {
    GatherArgs c[4];
};

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == DCV_ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (act == DCV_ACT_TANH) return tanhf(v);
    return v;
}

// straight-line (no scalar branches): the whole descriptor is read with wide scalar loads up front
__device__ __forceinline__ uint32_t dim_mask(const DimTaps t, int o, int shift) {
    uint32_t m = 0;
    const int p0 = o * t.mul + t.base;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = p0 + t.delta[u];
        m |= ((unsigned)p < (unsigned)t.size && u < t.n) ? (1u << (shift + u)) : 0u;
    }
    return m;
}


// --------------------------------------------------------------------------- //
// Epilogue of the gather GEMMs: scatter a wave's TOC x TM accumulator tiles to the strided output.
// Raw buffer stores: the per-lane voffset (position, + the lane's 4-row half) is computed once per
// 32-position column, the output channel rides on the scalar soffset, so a store costs one VMEM and
// one SALU instruction; lanes past M carry the out-of-range voffset and are dropped by the hardware.
// (The first version re-derived a 64-bit pointer and re-read its kernel arguments for every one of the
// 64 stores: 80k cycles per wave, a quarter of a workgroup's lifetime on the short-K layers.)
// --------------------------------------------------------------------------- //
template <int TOC, int TM, int ACT, bool GENERIC>
__device__ __forceinline__ void store_tiles(const f32x16 (&acc)[TOC][TM], const __amdgpu_buffer_rsrc_t yrs, const uint32_t (&voff)[TM],
                                            const int ocw, const int lhi, const int OC, const uint32_t y_sc4, const int act, const float slope,
                                            const bool accumulate, const __amdgpu_buffer_rsrc_t grs, const bool gated, const float gslope) {
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < TOC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ocr = i * 32 + (r & 3) + 8 * (r >> 2);
                const uint32_t soff = (uint32_t)(ocw + ocr) * y_sc4;
                float v = acc[i][j][r];
                uint32_t vo = voff[j];
                if constexpr (GENERIC) {
                    if (ocw + ocr + 4 * lhi >= OC) vo = 0x80000000u;
                    if (accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, vo, soff, 0));
                    if (gated) v *= __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, vo, soff, 0)) > 0.f ? 1.f : gslope;
                    v = apply_act(v, act, slope);
                } else {
                    if constexpr (ACT == DCV_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                    if constexpr (ACT == DCV_ACT_TANH) v = tanhf(v);
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yrs, vo, soff, 0);
            }
}

// per-lane byte offset of position m (relative to sample n0's output) + the lane's 4-row half
__device__ __forceinline__ uint32_t out_voffset(const GatherArgs& a, int m, uint32_t n0, int lhi, uint32_t y_sc4) {
    if (m >= a.M) return 0x80000000u;
    const uint32_t n = fdiv((uint32_t)m, a.div_sp);
    uint32_t r0 = (uint32_t)m - n * a.div_sp.div;
    const uint32_t od = fdiv(r0, a.div_hw);
    r0 -= od * a.div_hw.div;
    const uint32_t oh = fdiv(r0, a.div_w);
    const uint32_t ow = r0 - oh * a.div_w.div;
    return 4u * (uint32_t)((int)(n - n0) * (int)a.y_sn + (int)od * (int)a.y_sd + (int)oh * (int)a.y_sh + (int)ow * (int)a.y_sw) + (uint32_t)(4 * lhi) * y_sc4;
}

template <int TOC, int TM>
__device__ __forceinline__ void gather_epilogue(const GatherArgs& a, const f32x16 (&acc)[TOC][TM], int m0, int mcol0, int l31, int lhi, int ocw, uint32_t n0) {
    const int OC = a.OC, OCp = a.OCp, act = a.act;
    const bool accumulate = a.accumulate != 0;
    const float slope = a.slope;
    const uint32_t y_sc4 = (uint32_t)a.y_sc * 4u;
    // (p_pad2 = 0x5701, DCV_DEBUG_NOSTORE=1: an empty descriptor — every store is dropped by the range check; timing experiments only: what the epilogue's stores cost)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y + a.y_off + (int64_t)n0 * a.y_sn, 0, a.p_pad2 == 0x5701 ? 0u : 0x80000000u, 0x00020000);
    uint32_t voff[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) voff[j] = out_voffset(a, m0 + mcol0 + j * 32 + l31, n0, lhi, y_sc4);
    const bool gated = a.gate != nullptr;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gated ? a.gate : a.y) + a.y_off + (int64_t)n0 * a.y_sn, 0, 0x80000000u, 0x00020000);
    if (OC == OCp && !accumulate && !gated) {
        if (act == DCV_ACT_NONE) store_tiles<TOC, TM, DCV_ACT_NONE, false>(acc, yrs, voff, ocw, lhi, OC, y_sc4, act, slope, false, grs, false, 0.f);
        else if (act == DCV_ACT_LEAKY) store_tiles<TOC, TM, DCV_ACT_LEAKY, false>(acc, yrs, voff, ocw, lhi, OC, y_sc4, act, slope, false, grs, false, 0.f);
        else store_tiles<TOC, TM, DCV_ACT_TANH, false>(acc, yrs, voff, ocw, lhi, OC, y_sc4, act, slope, false, grs, false, 0.f);
    } else {
        store_tiles<TOC, TM, 0, true>(acc, yrs, voff, ocw, lhi, OC, y_sc4, act, slope, accumulate, grs, gated, a.gate_slope);
    }
}

// --------------------------------------------------------------------------- //
// gather GEMM.  Block = 256 threads = 4 waves laid out WOC x WM; each wave owns
// TOC x TM MFMA tiles of 32x32.  BN = 32*TOC*WOC output channels, BM = 32*TM*WM
// positions, K step 16.
// --------------------------------------------------------------------------- //
template <int TOC, int TM, int WOC, int WM, bool STRUCT>
__global__ __launch_bounds__(256, 3) void gather_gemm_kernel(const GatherArgs a) {
    constexpr int BN = 32 * TOC * WOC;
    constexpr int BM = 32 * TM * WM;
    constexpr int XPT = 16 * BM / 256;        // gathered elements per thread per K step
    static_assert(WOC * WM == 4, "4 waves");
    static_assert(BM == 64 || BM == 128 || BM == 256, "BM");
    constexpr int WF4 = 16 * BN / 4;          // float4s in one W tile
    constexpr int WPT = (WF4 + 255) / 256;

    __shared__ __attribute__((aligned(16))) float Xs[2][16][BM];
    __shared__ __attribute__((aligned(16))) float Ws[2][16][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int woc = wave / WM, wm = wave % WM;

    const int tiles_oc = a.OCp / BN;
    const int oc_t = blockIdx.x % tiles_oc;
    const int m_t = blockIdx.x / tiles_oc;
    const int oc0 = oc_t * BN;
    const int m0 = m_t * BM;

    // sample index of the block's first position: all 32-bit offsets are relative to it
    const uint32_t n0 = fdiv((uint32_t)m0, a.div_sp);
    // Gathered operand through a raw buffer descriptor: an element outside the tensor (padding) is
    // given the offset 0x80000000 >= num_records, for which the hardware returns 0 — no branch,
    // no select, and hipcc counts the loads so they stay in flight under the MFMAs.
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (int64_t)n0 * a.x_sn), 0, 0x80000000u, 0x00020000);

    // ---- this thread's gather column; K rows ksub*XPT .. ksub*XPT + XPT-1 of every 16-row step ----
    const int lm = tid % BM;
    const int ksub = __builtin_amdgcn_readfirstlane(tid / BM);
    int xbase4 = 0;
    uint32_t vmask = 0;
    {
        const int m = m0 + lm;
        if (m < a.M) {
            const uint32_t n = fdiv((uint32_t)m, a.div_sp);
            uint32_t r = (uint32_t)m - n * a.div_sp.div;
            const uint32_t od = fdiv(r, a.div_hw);
            r -= od * a.div_hw.div;
            const uint32_t oh = fdiv(r, a.div_w);
            const uint32_t ow = r - oh * a.div_w.div;
            vmask = dim_mask(a.td, (int)od, 0) | dim_mask(a.th, (int)oh, 8) | dim_mask(a.tw, (int)ow, 16);
            xbase4 = 4 * ((int)((int64_t)(n - n0) * a.x_sn) + ((int)od * a.td.mul + a.td.base) * a.x_sd +
                          ((int)oh * a.th.mul + a.th.base) * a.x_sh + ((int)ow * a.tw.mul + a.tw.base) * a.x_sw);
        }
    }

    f32x16 acc[TOC][TM];
#pragma unroll
    for (int i = 0; i < TOC; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float xv[XPT];
    float4 wv[WPT];

    // structured walk: per-thread voffsets fixed for the whole K loop (padding folded in as 0x80000000)
    uint32_t vloc[STRUCT ? XPT : 1];
    if constexpr (STRUCT) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int r = ksub * XPT + i;
            const uint32_t sel = a.s_sel[r];
            vloc[i] = ((vmask & sel) == sel) ? (uint32_t)(xbase4 + a.s_local[r]) : 0x80000000u;
        }
    }

    auto load_tile = [&](int it) {
        if constexpr (STRUCT) {
            const int soff = (it >> a.s_log2p) * a.s_stepA + (it & ((1 << a.s_log2p) - 1)) * a.s_stepD;
#pragma unroll
            for (int i = 0; i < XPT; ++i)
                xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vloc[i], soff, 0));
        } else {
        // this wave's XPT index rows: wave-uniform addresses -> wide scalar loads, issued first
        typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
        const i32x4* __restrict__ ko4 = reinterpret_cast<const i32x4*>(a.koff + it * 16 + ksub * XPT);
        const i32x4* __restrict__ ks4 = reinterpret_cast<const i32x4*>(a.ksel + it * 16 + ksub * XPT);
        int32_t ko[XPT];
        uint32_t ks[XPT];
#pragma unroll
        for (int q = 0; q < XPT / 4; ++q) {
            const i32x4 o = ko4[q], t = ks4[q];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ko[4 * q + c] = o[c];
                ks[4 * q + c] = (uint32_t)t[c];
            }
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            uint32_t vo = (uint32_t)(xbase4 + ko[i]);
            vo = ((vmask & ks[i]) == ks[i]) ? vo : 0x80000000u;
            xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vo, 0, 0));
        }
        }
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int f = tid + 256 * j;
            if (WF4 % 256 == 0 || f < WF4) {
                const int row = f / (BN / 4), c4 = f % (BN / 4);
                wv[j] = *reinterpret_cast<const float4*>(a.wp + (int64_t)(it * 16 + row) * a.OCp + oc0 + c4 * 4);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) Xs[buf][ksub * XPT + i][lm] = xv[i];
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int f = tid + 256 * j;
            if (WF4 % 256 == 0 || f < WF4) {
                const int row = f / (BN / 4), c4 = f % (BN / 4);
                *reinterpret_cast<float4*>(&Ws[buf][row][c4 * 4]) = wv[j];
            }
        }
    };

    const int it0 = a.slab ? blockIdx.y * a.kper : 0;
    const int it1 = a.slab ? min(a.KIT, it0 + a.kper) : a.KIT;
    load_tile(it0);
    store_tile(0);
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
#ifdef DCV_STAMP
    unsigned long long seg0 = 0, seg1 = 0, seg2 = 0, seg3 = 0;
    const unsigned long long tstart = clock64();
#endif
    for (int it = it0; it < it1; ++it) {
        const int buf = (it - it0) & 1;
        __builtin_amdgcn_sched_barrier(0);
        STAMP(t0);
        if (it + 1 < it1) load_tile(it + 1);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(t1);
        // a wave that has entered its MFMA phase outranks the co-resident waves (other blocks) that are
        // still issuing loads/stores: phases rotate instead of interleaving into a convoy
        __builtin_amdgcn_s_setprio(2);
        // fragments of k-step ks+1 are read from LDS while the MFMAs of step ks run
        float af[2][TOC], bf[2][TM];
#pragma unroll
        for (int i = 0; i < TOC; ++i) af[0][i] = Ws[buf][lhi][(woc * TOC + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < TM; ++j) bf[0][j] = Xs[buf][lhi][(wm * TM + j) * 32 + l31];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 8) {
                const int k = 2 * (ks + 1) + lhi;
#pragma unroll
                for (int i = 0; i < TOC; ++i) af[nxt][i] = Ws[buf][k][(woc * TOC + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < TM; ++j) bf[nxt][j] = Xs[buf][k][(wm * TM + j) * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < TOC; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            // pin the order hipcc would otherwise undo: next step's LDS reads, THEN this step's MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, TOC + TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TOC * TM, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(t2);
        if (it + 1 < it1) store_tile(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(t3);
        __syncthreads();
#ifdef DCV_STAMP
        { __builtin_amdgcn_sched_barrier(0); unsigned long long t4 = clock64(); seg0 += t1 - t0; seg1 += t2 - t1; seg2 += t3 - t2; seg3 += t4 - t3; }
#endif
    }
#ifdef DCV_STAMP
    if (lane == 0 && blockIdx.x < 4096) {
        unsigned long long* g = g_stamp[blockIdx.x][wave];
        g[0] = seg0; g[1] = seg1; g[2] = seg2; g[3] = seg3; g[4] = clock64() - tstart; g[5] = (unsigned long long)(it1 - it0);
    }
#endif

    if (a.slab) {  // raw partial sums, GEMM layout, padded so no bounds checks
        float* __restrict__ sl = a.slab + (int64_t)blockIdx.y * a.OCp * a.Mp;
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TOC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int oc = oc0 + (woc * TOC + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    sl[(int64_t)oc * a.Mp + m0 + (wm * TM + j) * 32 + l31] = acc[i][j][r];
                }
        return;
    }

    // ---- epilogue: acc[i][j][r] -> oc = .. + (r&3) + 8*(r>>2) + 4*lhi, m = .. + l31 ----
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + (wm * TM + j) * 32 + l31;
        if (m >= a.M) continue;
        const uint32_t n = fdiv((uint32_t)m, a.div_sp);
        uint32_t r0 = (uint32_t)m - n * a.div_sp.div;
        const uint32_t od = fdiv(r0, a.div_hw);
        r0 -= od * a.div_hw.div;
        const uint32_t oh = fdiv(r0, a.div_w);
        const uint32_t ow = r0 - oh * a.div_w.div;
        float* __restrict__ yb = a.y + a.y_off + (int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)oh * a.y_sh + (int64_t)ow * a.y_sw;
#pragma unroll
        for (int i = 0; i < TOC; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = oc0 + (woc * TOC + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (oc < a.OC) {
                    float v = acc[i][j][r];
                    float* p = yb + (int64_t)oc * a.y_sc;
                    if (a.accumulate) v += *p;
                    if (a.gate) v *= a.gate[p - a.y] > 0.f ? 1.f : a.gate_slope;
                    *p = apply_act(v, a.act, a.slope);
                }
            }
        }
    }
}

// --------------------------------------------------------------------------- //
// gather GEMM, LDS-DMA form (structured K walk only).  Same tiling and epilogue as
// gather_gemm_kernel, but both operand tiles go global -> LDS directly
// (buffer_load ... lds): the X tile row k / 64 consecutive positions of a wave is exactly one
// lane-linear 256-B DMA write, the packed W tile is lane-linear in float4s.  No staging
// registers, no LDS-store phase, one barrier per K step; padding still arrives as zeros through
// the out-of-range voffset.  Per step: wait own DMAs + barrier, issue next tile's DMAs into the
// other buffer, 32 MFMAs per wave on this one.
// --------------------------------------------------------------------------- //
// Per-tile BatchNorm partial sums from the accumulators (positions past M hold exact zeros: their operand
// rows were padding).  Rows of one wave are reduced over its 32-lane halves with xor-shuffles, the WM waves
// that share an output-channel row meet in LDS and are added in a fixed order.
// Sum over each 32-lane half of the wave, valid in lanes 16-31 / 48-63: five DPP adds (quad swaps, half-row and
// row mirrors, then lane 15 of the even rows broadcast into the odd rows) instead of five LDS-crossbar shuffles.
__device__ __forceinline__ float half_wave_sum(float v) {
    auto dpp = [](float x, auto ctrl, auto rows) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rows)::value, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});   // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});   // row_mirror
    v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1 and 3
    return v;
}

template <int TOC, int TM, int WOC, int WM>
__device__ __forceinline__ void tile_bn_partials(const GatherArgs& a, const f32x16 (&acc)[TOC][TM], float* smem, int tid, int lane, int woc, int wm, int oc0, int m_t) {
    constexpr int BN = 32 * TOC * WOC;
    const int l31 = lane & 31, lhi = lane >> 5;
    __syncthreads();   // every wave has left the K loop: the tile buffers are free
#pragma unroll
    for (int i = 0; i < TOC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < TM; ++j) { const float v = acc[i][j][r]; s1 += v; s2 += v * v; }
            s1 = half_wave_sum(s1);
            s2 = half_wave_sum(s2);
            if (l31 == 31) {
                const int ocl = (woc * TOC + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                smem[(wm * BN + ocl) * 2] = s1;
                smem[(wm * BN + ocl) * 2 + 1] = s2;
            }
        }
    __syncthreads();
    float* __restrict__ dst = a.stat + ((int64_t)(a.stat_cls * a.stat_ntm + m_t) * a.OCp + oc0) * 2;
    for (int e = tid; e < 2 * BN; e += 256) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) t += smem[w * BN * 2 + e];
        dst[e] = t;
    }
}

typedef __attribute__((address_space(3))) void lds_void;

// DSTEP (structured == 2): a K step is (4 channels, ONE depth tap, 2x2 inner taps) — the 3-D discriminators'
// data gradients, where 20-30 % of the (position, depth tap) pairs are padding.  Steps whose depth tap is
// outside the tensor for every position of the tile are skipped outright; for the rest the tap's validity
// is OR-ed into the per-lane voffsets (one VALU op per DMA).
// BF: 0 = fp32 MFMA, 1 = bf16 products, 2 = fp32 on the bf16 pipe (three bf16 pieces per operand, six products; split3_bf16x8)
template <int TOC, int TM, int WOC, int WM, bool DSTEP, bool PATCH, int BF = 0>
__global__ __launch_bounds__(256, (BF == 2 ? 3 : 4)) void gather_gemm_dma_kernel(const GatherArgsPack pack) {
    constexpr int BN = 32 * TOC * WOC;
    constexpr int BM = 32 * TM * WM;
    // XCD-aware workgroup -> tile mapping.  Workgroup ids go round-robin over the 8 XCDs, each with its own
    // L2: the workgroups that read the SAME gathered operand (the op's stride-parity classes and the
    // output-channel tiles of one position tile) are given ids 8 apart, i.e. the same XCD back to back, so
    // the operand is fetched into one L2 once instead of into up to 8 of them at different times.
    const unsigned tiles_oc_ = (unsigned)pack.c[0].OCp / BN, ncls_ = (unsigned)pack.c[0].pad0;
    const unsigned grp_ = tiles_oc_ * ncls_, loc_ = blockIdx.x >> 3;
    const unsigned g_ = loc_ % grp_;
    const int m_t = (int)((loc_ / grp_) * 8 + (blockIdx.x & 7));
    const int oc_t = (int)(g_ % tiles_oc_);
    const GatherArgs& a = pack.c[g_ / tiles_oc_];
#ifdef DCV_STAMP
    const unsigned long long q_start = clock64();
    unsigned long long q_wait = 0, q_loop = 0;
#endif
    constexpr int XPT = 16 * BM / 256;
    // W tile of a K step in 16-byte granules: fp32 [16][BN] floats; bf16 products: packed bf16 [2][BN][8] (PackArgs.fmt 1), half the bytes;
    // fp32 on the bf16 pipe: three such planes [3][2][BN][8] (PackArgs.fmt 2)
    constexpr int WF4 = BF == 2 ? 6 * BN : BF ? 2 * BN : 16 * BN / 4;
    constexpr int WPT = (WF4 + 255) / 256;
    constexpr int WSZ = BF == 2 ? 24 * BN : 16 * BN;   // floats per W buffer
    static_assert(WOC * WM == 4, "4 waves");
    // ONE LDS array: [2][16][BM] X tiles then [2][WSZ] W tiles
    __shared__ __attribute__((aligned(16))) float smem[2 * 16 * BM + 2 * WSZ];
    float* const Xs = smem;
    float* const Ws = smem + 2 * 16 * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int woc = wave / WM, wm = wave % WM;
    const int oc0 = oc_t * BN;
    const int m0 = m_t * BM;
    if (m0 >= a.Mp) return;   // grid padding; classes of one launch can differ by a row/column of positions
    const bool slabmode = a.slab != nullptr && m0 >= a.rag_m0;
    if (!slabmode && blockIdx.y != 0) return;   // ragged split-K: a directly stored tile exists once

    const uint32_t n0 = fdiv((uint32_t)m0, a.div_sp);
    // DSTEP: the scalar depth-tap offset counts up from the farthest tap, so the base sits x_back elements
    // before the sample (addresses below the tensor are only ever formed for padding, which is not read)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (int64_t)n0 * a.x_sn - (DSTEP ? a.x_back : 0)), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wp), 0, 0x80000000u, 0x00020000);

    const int lm = tid % BM;
    const int ksub = __builtin_amdgcn_readfirstlane(tid / BM);
    const int lm_wave = __builtin_amdgcn_readfirstlane(lm - lane);   // first position column of this wave
    const int l31 = lane & 31, lhi = lane >> 5;
    constexpr int NS = XPT / 4;   // patch form: 16-byte DMA slots per wave and step (the X buffer holds 4 * BM granules)
    uint32_t vmask = 0;           // gathered form: this lane's position; patch form: OR of the slots' depth bits
    uint32_t vloc[PATCH ? NS : XPT];
    uint32_t dmrow[PATCH ? NS : 1];
    uint32_t fb[PATCH ? TM : 1];
    if constexpr (PATCH) {
        const uint32_t GPR = a.p_gpr.div;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            // granule q of the step's patch: ((channel * TR + output row) * NH + input row) * GPR + column granule
            const uint32_t q = (uint32_t)((sl * 4 + wave) * 64 + lane);
            const uint32_t q1 = fdiv(q, a.p_gpr);
            const int c = (int)(q - q1 * GPR) - 1;                      // input columns 4c .. 4c + 3 (c = -1, IW/4: zero halo)
            const int khp = (int)(q1 & ((1u << a.p_log2nh) - 1));
            const uint32_t q2 = q1 >> a.p_log2nh;
            const int t = (int)(q2 & ((1u << a.p_log2tr) - 1));
            const int ch = (int)(q2 >> a.p_log2tr);
            const int m = m0 + (t << a.p_log2ow);
            const uint32_t mm = (uint32_t)(m < a.M ? m : m0);
            const uint32_t n = fdiv(mm, a.div_sp);
            uint32_t r = mm - n * a.div_sp.div;
            const uint32_t od = fdiv(r, a.div_hw);
            r -= od * a.div_hw.div;
            const uint32_t oh = fdiv(r, a.div_w);
            const int ih = (int)oh * a.th.mul + a.p_ihmin + khp;
            const bool ok = (int)q < a.p_G && m < a.M && (unsigned)ih < (unsigned)a.th.size && (unsigned)(4 * c) < (unsigned)a.tw.size;
            vloc[sl] = ok ? (uint32_t)(4 * ((int)((int64_t)(n - n0) * a.x_sn) + ((int)od * a.td.mul + a.td.base) * a.x_sd + ih * a.x_sh + 4 * c) + ch * a.p_sc4)
                          : 0x80000000u;
            dmrow[sl] = ok ? dim_mask(a.td, (int)od, 0) : 0u;
            vmask |= dmrow[sl];
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int col = (wm * TM + j) * 32 + l31;
            const int t = col >> a.p_log2ow, ow = col & ((1 << a.p_log2ow) - 1);
            fb[j] = (uint32_t)(4 * (((t << a.p_log2nh) * a.p_iwp) + ow * a.tw.mul + a.p_iwmin4 + lhi * a.p_dw1));
        }
    } else {
    int xbase4 = 0;
    {
        const int m = m0 + lm;
        if (m < a.M) {
            const uint32_t n = fdiv((uint32_t)m, a.div_sp);
            uint32_t r = (uint32_t)m - n * a.div_sp.div;
            const uint32_t od = fdiv(r, a.div_hw);
            r -= od * a.div_hw.div;
            const uint32_t oh = fdiv(r, a.div_w);
            const uint32_t ow = r - oh * a.div_w.div;
            vmask = dim_mask(a.td, (int)od, 0) | dim_mask(a.th, (int)oh, 8) | dim_mask(a.tw, (int)ow, 16);
            xbase4 = 4 * ((int)((int64_t)(n - n0) * a.x_sn) + ((int)od * a.td.mul + a.td.base) * a.x_sd +
                          ((int)oh * a.th.mul + a.th.base) * a.x_sh + ((int)ow * a.tw.mul + a.tw.base) * a.x_sw);
        }
    }
    {
        // this wave's XPT table rows: two wide scalar loads instead of 2 * XPT single ones, each with its wait
        typedef int32_t i32xp __attribute__((ext_vector_type(XPT)));
        const i32xp sl = *reinterpret_cast<const i32xp*>(a.s_local + ksub * XPT);
        const i32xp ss = *reinterpret_cast<const i32xp*>(a.s_sel + ksub * XPT);
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const uint32_t sel = (uint32_t)ss[i];
            vloc[i] = ((vmask & sel) == sel) ? (uint32_t)(xbase4 + sl[i]) : 0x80000000u;
        }
    }
    dmrow[0] = 0;
    fb[0] = 0;
    }
    // patch form: byte offset of k rows 0, 2, .. 14 of a step inside the patch (odd rows: one tap further, in fb)
    int32_t koff[8];
    {
        typedef int32_t i32x16 __attribute__((ext_vector_type(16)));
        const i32x16 sl16 = *reinterpret_cast<const i32x16*>(a.s_local);
#pragma unroll
        for (int k = 0; k < 8; ++k) koff[k] = PATCH ? sl16[2 * k] : 0;
    }
    // bf16 products: a lane of half-wave lhi holds k rows 8 lhi .. 8 lhi + 7 of the step; patch form: their byte offsets
    // inside the patch (the fp32 form's fb[] carries the odd-row tap step, which is taken back out here)
    uint32_t koff8[BF && PATCH ? 8 : 1], fbb[BF && PATCH ? TM : 1];
    if constexpr (BF && PATCH) {
        typedef int32_t i32x16b __attribute__((ext_vector_type(16)));
        const i32x16b slb = *reinterpret_cast<const i32x16b*>(a.s_local);
#pragma unroll
        for (int q = 0; q < 8; ++q) koff8[q] = (uint32_t)(lhi ? slb[8 + q] : slb[q]);
#pragma unroll
        for (int j = 0; j < TM; ++j) fbb[j] = fb[j] - (uint32_t)(4 * lhi * a.p_dw1);
    } else {
        koff8[0] = 0; fbb[0] = 0;
    }
    // W tile: float4 index f = tid + 256 j -> row f / (BN/4), column 4 (f % (BN/4)); LDS offset = 4 f floats
    uint32_t wvo[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int f = tid + 256 * j;
        if constexpr (BF) {   // granule f = (k block f / BN, output channel f % BN)
            wvo[j] = (uint32_t)(16 * ((f / BN) * a.OCp + oc0 + f % BN));
        } else {
            const int row = f / (BN / 4), c4 = f % (BN / 4);
            wvo[j] = (uint32_t)(4 * (row * a.OCp + oc0 + c4 * 4));
        }
    }
    const int wstep4 = BF == 2 ? 6 * a.OCp * 16 : BF ? 2 * a.OCp * 16 : 16 * a.OCp * 4;

#ifdef DCV_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long q_a = clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
    // DSTEP: depth tap of step `it` is ud = nd-1 - (it & (nd-1)) (ascending addresses); its validity bit
    // for this lane's position is bit ud of vmask; an invalid tap turns every voffset of the step into padding
    const int ndm1 = (1 << a.s_log2p) - 1;
#define DCV_DFLAG(IT) (DSTEP ? ((((vmask >> (ndm1 - ((IT) & ndm1))) & 1u) ^ 1u) << 31) : 0u)
#define DCV_DFLAG_SLOT(IT, S) (DSTEP ? ((((dmrow[S] >> (ndm1 - ((IT) & ndm1))) & 1u) ^ 1u) << 31) : 0u)
    uint32_t dmask = 0xffu;   // depth taps that any position of the tile can use
    if constexpr (DSTEP) {   // bitwise OR over the block (word 0 of the tile memory as scratch, before any DMA)
        uint32_t wm = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (__ballot((vmask >> u) & 1u)) wm |= 1u << u;
        uint32_t* sc = reinterpret_cast<uint32_t*>(smem);
        if (tid == 0) *sc = 0;
        __syncthreads();
        if (lane == 0) atomicOr(sc, wm);
        __syncthreads();
        dmask = *sc;
        __syncthreads();
    }
#define DCV_STEP_LIVE(IT) ((dmask >> (ndm1 - ((IT) & ndm1))) & 1u)
#ifdef DCV_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long q_b = clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif

    // The host pass type-checks builtins without gfx950 target features and rejects the 16-byte
    // LDS-DMA size, which silently drops the kernel's host stub: device pass only.
#if defined(__HIP_DEVICE_COMPILE__)
// one X DMA: gathered form = k row ksub*XPT + I (dword per lane); patch form = granule slot I (16 bytes per lane)
#define DCV_ISSUE_X(IT, SOFF, BUF, I)                                                                                   \
    {                                                                                                                   \
        if constexpr (PATCH)                                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(Xs + (BUF) * 16 * BM + (((I) * 4 + wave) * 64) * 4), 16, \
                                                     vloc[(I) < NS ? (I) : 0] | DCV_DFLAG_SLOT(IT, (I) < NS ? (I) : 0), (SOFF), 0, 0); \
        else                                                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(Xs + (BUF) * 16 * BM + lm_wave + (ksub * XPT + (I)) * BM), 4, \
                                                     vloc[I] | DCV_DFLAG(IT), (SOFF), 0, 0);                            \
    }
#define DCV_ISSUE_TILE(IT, BUF)                                                                                         \
    {                                                                                                                   \
        const int it_ = (IT);                                                                                           \
        const int soff_ = (it_ >> a.s_log2p) * a.s_stepA + (it_ & ((1 << a.s_log2p) - 1)) * a.s_stepD;                  \
        _Pragma("unroll") for (int i = 0; i < (PATCH ? NS : XPT); ++i) DCV_ISSUE_X(it_, soff_, BUF, i)                  \
        float* wb_ = Ws + (BUF) * WSZ;                                                                                  \
        _Pragma("unroll") for (int j = 0; j < WPT; ++j)                                                                 \
            if (WF4 % 256 == 0 || wave * 64 + 256 * j < WF4) /* wave-uniform: whole waves only */                       \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(wb_ + (wave * 64 + 256 * j) * 4), 16, wvo[j], it_ * wstep4, 0, 0); \
    }
#define DCV_ISSUE_W(IT, BUF)                                                                                            \
    {                                                                                                                   \
        float* wb_ = Ws + (BUF) * WSZ;                                                                                  \
        _Pragma("unroll") for (int j = 0; j < WPT; ++j)                                                                 \
            if (WF4 % 256 == 0 || wave * 64 + 256 * j < WF4)                                                            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void*)(wb_ + (wave * 64 + 256 * j) * 4), 16, wvo[j], (IT) * wstep4, 0, 0); \
    }
#else
#define DCV_ISSUE_TILE(IT, BUF) { (void)wstep4; (void)wvo; (void)vloc; (void)lm_wave; (void)dmrow; }
#define DCV_ISSUE_X(IT, SOFF, BUF, I) { (void)(SOFF); }
#define DCV_ISSUE_W(IT, BUF) { (void)wstep4; (void)wvo; }
#endif

    f32x16 acc[TOC][TM];
#pragma unroll
    for (int i = 0; i < TOC; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int it0 = slabmode ? blockIdx.y * a.kper : 0;
    const int it1 = slabmode ? min(a.KIT, it0 + a.kper) : a.KIT;
    const int nst = it1 - it0;
#define DCV_IT(J) (it0 + (J))
    int j0 = 0;
    if constexpr (DSTEP)
        while (j0 < nst && !DCV_STEP_LIVE(DCV_IT(j0))) ++j0;
    if (j0 < nst) DCV_ISSUE_TILE(DCV_IT(j0), 0)
    int buf = 0;
#ifdef DCV_STAMP
    const unsigned long long q_pro = clock64();
#endif
    [[maybe_unused]] bool x6_neg = false;   // BF == 2: the accumulators currently hold minus the running sum
    for (int j = j0; j < nst; buf ^= 1) {
        [[maybe_unused]] const int it_cur = DCV_IT(j);
        int nx = j + 1;
        if constexpr (DSTEP)
            while (nx < nst && !DCV_STEP_LIVE(DCV_IT(nx))) ++nx;
        STAMP(q0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMAs of the current tile have landed
        __syncthreads();                                   // ... everyone's have; and all reads of buf^1 are done
        STAMP(q1);
        // next tile's DMAs (clamped on the last step: a harmless repeat into the idle buffer) are
        // spread over the k-steps below, two per MFMA group, so their issue cost hides under the MFMAs
        __builtin_amdgcn_s_setprio(2);
        const int itn = DCV_IT(nx < nst ? nx : j);   // last step: a harmless repeat into the idle buffer
        const int soffn = (itn >> a.s_log2p) * a.s_stepA + (itn & ((1 << a.s_log2p) - 1)) * a.s_stepD;
        j = nx;
        DCV_ISSUE_W(itn, buf ^ 1)
        const float* xt = Xs + buf * 16 * BM;
        const float* wt = Ws + buf * WSZ;
        if constexpr (BF == 2) {
            // fp32 on the bf16 pipe: the weights arrive pre-split (three 16-byte LDS reads per A fragment), the activations are split here;
            // six v_mfma_f32_32x32x16_bf16 per (i, j) cover the step's 16 k rows
#pragma unroll
            for (int i = 0; i < (PATCH ? NS : XPT); ++i) DCV_ISSUE_X(itn, soffn, buf ^ 1, i)
            bf16x8 a8[TOC][3], b8[TM][3];
#pragma unroll
            for (int i = 0; i < TOC; ++i)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    a8[i][pc] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(wt) + 16 * ((pc * 2 + lhi) * BN + (woc * TOC + i) * 32 + l31));
            float tb[TM][8];
#pragma unroll
            for (int jj = 0; jj < TM; ++jj)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    tb[jj][q] = PATCH ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xt) + (fbb[PATCH ? jj : 0] + koff8[PATCH ? q : 0]))
                                      : xt[(8 * lhi + q) * BM + (wm * TM + jj) * 32 + l31];
            // position column jj + 1 is split (44 VALU operations) in the shadow of column jj's 6 TOC MFMAs: an MFMA holds the matrix pipe for 32
            // cycles and the vector issue for 8 of them, so ~4 operations behind each one issue for free; only column 0's split is exposed
            split3_bf16x8(tb[0], b8[0]);
            if (X6_PHASE(it_cur) != x6_neg) {   // phase change (wave-uniform): flip the accumulators' sign
                x6_neg = !x6_neg;
#pragma unroll
                for (int i = 0; i < TOC; ++i)
#pragma unroll
                    for (int jj = 0; jj < TM; ++jj)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][jj][r] = -acc[i][jj][r];
            }
#pragma unroll
            for (int jj = 0; jj < TM; ++jj) {
                if (jj + 1 < TM) split3_bf16x8(tb[jj + 1 < TM ? jj + 1 : 0], b8[jj + 1 < TM ? jj + 1 : 0]);
#pragma unroll
                for (int i = 0; i < TOC; ++i) DCV_MFMA_X6(acc[i][jj], a8[i], b8[jj])
                if (jj + 1 < TM) {
#pragma unroll
                    for (int q = 0; q < 6 * TOC; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            continue;
        } else if constexpr (BF == 1) {
            // one v_mfma_f32_32x32x16_bf16 per (i, j) covers the step's 16 k rows: the next tile's X DMAs go out first,
            // then 8 LDS reads + 4 packed conversions per fragment
#pragma unroll
            for (int i = 0; i < (PATCH ? NS : XPT); ++i) DCV_ISSUE_X(itn, soffn, buf ^ 1, i)
            bf16x8 a8[TOC], b8[TM];
#pragma unroll
            for (int i = 0; i < TOC; ++i)   // packed bf16 weights: k rows 8 lhi .. 8 lhi + 7 of output channel row l31 are ONE 16-byte LDS read
                a8[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(wt) + 16 * (lhi * BN + (woc * TOC + i) * 32 + l31));
#pragma unroll
            for (int jj = 0; jj < TM; ++jj) {
                float t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    t[q] = PATCH ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xt) + (fbb[PATCH ? jj : 0] + koff8[PATCH ? q : 0]))
                                 : xt[(8 * lhi + q) * BM + (wm * TM + jj) * 32 + l31];
                b8[jj] = pack_bf16x8(t);
            }
#pragma unroll
            for (int i = 0; i < TOC; ++i)
#pragma unroll
                for (int jj = 0; jj < TM; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[jj], acc[i][jj], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            continue;
        }
        float af[2][TOC], bf[2][TM];
#pragma unroll
        for (int i = 0; i < TOC; ++i) af[0][i] = wt[lhi * BN + (woc * TOC + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < TM; ++j)
            bf[0][j] = PATCH ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xt) + (fb[PATCH ? j : 0] + (uint32_t)koff[0]))
                             : xt[lhi * BM + (wm * TM + j) * 32 + l31];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 8) {
                const int k = 2 * (ks + 1) + lhi;
#pragma unroll
                for (int i = 0; i < TOC; ++i) af[nxt][i] = wt[k * BN + (woc * TOC + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    bf[nxt][j] = PATCH ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xt) + (fb[PATCH ? j : 0] + (uint32_t)koff[ks + 1 < 8 ? ks + 1 : 0]))
                                       : xt[k * BM + (wm * TM + j) * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < TOC; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            // next tile's X DMAs in the shadow of this step's MFMAs: gathered form XPT/8 rows per k-step,
            // patch form one granule slot in each of the first NS k-steps
            if constexpr (PATCH) {
                if (ks < NS) DCV_ISSUE_X(itn, soffn, buf ^ 1, ks)
            } else {
                if constexpr (XPT >= 8) {
#pragma unroll
                    for (int q = 0; q < XPT / 8; ++q) DCV_ISSUE_X(itn, soffn, buf ^ 1, ks * (XPT / 8) + q)
                } else {   // 64-position tiles: XPT = 4 rows per wave, one in every other k-step
                    if ((ks & 1) == 0) DCV_ISSUE_X(itn, soffn, buf ^ 1, (ks / 2) < XPT ? ks / 2 : 0)
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, TOC + TM, 0);   // next step's fragments first,
            __builtin_amdgcn_sched_group_barrier(0x008, TOC * TM, 0);   // then this step's MFMAs,
            if (PATCH ? ks < NS : (XPT >= 8 || (ks & 1) == 0)) __builtin_amdgcn_sched_group_barrier(0x010, PATCH ? 1 : (XPT >= 8 ? XPT / 8 : 1), 0);   // then the DMA issues in their shadow
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef DCV_STAMP
        { unsigned long long q2 = clock64(); q_wait += q1 - q0; q_loop += q2 - q1; }
#endif
    }
#ifdef DCV_STAMP
    const unsigned long long q_fw = clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
    if constexpr (BF == 2) {
        if (x6_neg) {
#pragma unroll
            for (int i = 0; i < TOC; ++i)
#pragma unroll
                for (int jj = 0; jj < TM; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jj][r] = -acc[i][jj][r];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeated last-step DMAs must land before LDS is released
#ifdef DCV_STAMP
    const unsigned long long q_epi = clock64();
#define DCV_STAMP_OUT()                                                                                      \
    if (lane == 0) {                                                                                         \
        const unsigned bid_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                \
        if (bid_ < 4096) {                                                                                   \
            unsigned long long* g = g_stamp[bid_][wave];                                                     \
            const unsigned long long q_end = clock64();                                                      \
            g[0] = q_pro - q_start; g[1] = q_wait; g[2] = q_loop; g[3] = (q_end - q_epi) | ((q_epi - q_fw) << 32); g[4] = q_end - q_start; g[5] = (unsigned long long)nst | ((q_a - q_start) << 16) | ((q_b - q_a) << 40); \
        }                                                                                                    \
    }
#else
#define DCV_STAMP_OUT()
#endif

    if (slabmode) {
        float* __restrict__ sl = a.slab + (int64_t)blockIdx.y * a.OCp * a.slab_mp;
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TOC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int oc = oc0 + (woc * TOC + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    sl[(int64_t)oc * a.slab_mp + (m0 - a.rag_m0) + (wm * TM + j) * 32 + l31] = acc[i][j][r];
                }
        DCV_STAMP_OUT()
        return;
    }
    gather_epilogue<TOC, TM>(a, acc, m0, wm * TM * 32, l31, lhi, oc0 + woc * TOC * 32, n0);
    if (a.stat) tile_bn_partials<TOC, TM, WOC, WM>(a, acc, smem, tid, lane, woc, wm, oc0, m_t);
    DCV_STAMP_OUT()
}

// y = act( sum_s slab[s][oc][m] (+ y) ), scattered to the NCDHW output; fixed summation order.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GatherArgsPack pack, int S) {
    // 4 consecutive positions per thread (16-byte slab loads), 4 slabs in flight; 32-bit index math
    const GatherArgs& a = pack.c[blockIdx.y];
    const uint32_t Mp = (uint32_t)a.slab_mp, M = (uint32_t)a.M, q4 = Mp >> 2;   // the slab's own pitch: the whole op, or its ragged tail from rag_m0 on
    if (q4 == 0) return;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t oc = idx / q4, ml = (idx - oc * q4) * 4, m = ml + (uint32_t)a.rag_m0;
    if (oc >= (uint32_t)a.OC || m >= M) return;
    const float4* __restrict__ p = reinterpret_cast<const float4*>(a.slab + (int64_t)oc * Mp + ml);
    const int64_t stride = ((int64_t)a.OCp * Mp) >> 2;
    float4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
    int k = 0;
    for (; k + 3 < S; k += 4) {
        const float4 t0 = p[(int64_t)k * stride], t1 = p[(int64_t)(k + 1) * stride], t2 = p[(int64_t)(k + 2) * stride], t3 = p[(int64_t)(k + 3) * stride];
        v0.x += t0.x; v0.y += t0.y; v0.z += t0.z; v0.w += t0.w;
        v1.x += t1.x; v1.y += t1.y; v1.z += t1.z; v1.w += t1.w;
        v2.x += t2.x; v2.y += t2.y; v2.z += t2.z; v2.w += t2.w;
        v3.x += t3.x; v3.y += t3.y; v3.z += t3.z; v3.w += t3.w;
    }
    for (; k < S; ++k) {
        const float4 t0 = p[(int64_t)k * stride];
        v0.x += t0.x; v0.y += t0.y; v0.z += t0.z; v0.w += t0.w;
    }
    const float v[4] = {(v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y), (v0.z + v1.z) + (v2.z + v3.z), (v0.w + v1.w) + (v2.w + v3.w)};
    const int act = a.act, accumulate = a.accumulate;
    const float slope = a.slope;
    float* __restrict__ yb = a.y + a.y_off + (int64_t)oc * a.y_sc;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t me = m + e;
        if (me >= M) break;
        const uint32_t n = fdiv(me, a.div_sp);
        uint32_t r0 = me - n * a.div_sp.div;
        const uint32_t od = fdiv(r0, a.div_hw);
        r0 -= od * a.div_hw.div;
        const uint32_t oh = fdiv(r0, a.div_w);
        const uint32_t ow = r0 - oh * a.div_w.div;
        float* q = yb + (int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)oh * a.y_sh + (int64_t)ow * a.y_sw;
        float x = v[e];
        if (accumulate) x += *q;
        if (a.gate) x *= a.gate[q - a.y] > 0.f ? 1.f : a.gate_slope;
        *q = apply_act(x, act, slope);
    }
}

// --------------------------------------------------------------------------- //
// thin gather: OC <= 4 (RGB / depth / flow heads and stems' data gradients, D logits).
// A 32-wide MFMA tile would waste >= 7/8 of the matrix pipe on these, so they run as direct
// FMAs: 64 positions per block (lanes along m, coalesced), the block's 4 waves split the K
// range, weights Wp[k][4] and the index rows are wave-uniform scalar loads.  With a slab
// (blockIdx.y = K split) the partial sums go through splitk_reduce_kernel.
// --------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void thin_gather_kernel(const GatherArgs a) {
    __shared__ float red[3][4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 64;
    const uint32_t n0 = fdiv((uint32_t)m0, a.div_sp);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (int64_t)n0 * a.x_sn), 0, 0x80000000u, 0x00020000);
    const int m = m0 + lane;
    int xbase4 = 0;
    uint32_t vmask = 0;
    uint32_t n = 0, od = 0, oh = 0, ow = 0;
    if (m < a.M) {
        n = fdiv((uint32_t)m, a.div_sp);
        uint32_t r = (uint32_t)m - n * a.div_sp.div;
        od = fdiv(r, a.div_hw);
        r -= od * a.div_hw.div;
        oh = fdiv(r, a.div_w);
        ow = r - oh * a.div_w.div;
        vmask = dim_mask(a.td, (int)od, 0) | dim_mask(a.th, (int)oh, 8) | dim_mask(a.tw, (int)ow, 16);
        xbase4 = 4 * ((int)((int64_t)(n - n0) * a.x_sn) + ((int)od * a.td.mul + a.td.base) * a.x_sd +
                      ((int)oh * a.th.mul + a.th.base) * a.x_sh + ((int)ow * a.tw.mul + a.tw.base) * a.x_sw);
    }
    // this block's K rows (groups of 4), then this wave's share of them
    const int g_all = a.KIT * 4;
    const int gb0 = a.slab ? blockIdx.y * a.kper * 4 : 0;
    const int gb1 = a.slab ? min(g_all, gb0 + a.kper * 4) : g_all;
    const int gper = (gb1 - gb0 + 3) / 4;
    const int g0 = gb0 + wave * gper, g1 = min(gb1, g0 + gper);
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int g = g0; g < g1; ++g) {
        const i32x4 ko = *reinterpret_cast<const i32x4*>(a.koff + 4 * g);
        const i32x4 ks = *reinterpret_cast<const i32x4*>(a.ksel + 4 * g);
        float xv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t vo = (uint32_t)(xbase4 + ko[c]);
            vo = ((vmask & (uint32_t)ks[c]) == (uint32_t)ks[c]) ? vo : 0x80000000u;
            xv[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vo, 0, 0));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 w = *reinterpret_cast<const float4*>(a.wp + (int64_t)(4 * g + c) * 4);
            acc0 += w.x * xv[c]; acc1 += w.y * xv[c]; acc2 += w.z * xv[c]; acc3 += w.w * xv[c];
        }
    }
    if (wave > 0) { red[wave - 1][0][lane] = acc0; red[wave - 1][1][lane] = acc1; red[wave - 1][2][lane] = acc2; red[wave - 1][3][lane] = acc3; }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) { acc0 += red[w][0][lane]; acc1 += red[w][1][lane]; acc2 += red[w][2][lane]; acc3 += red[w][3][lane]; }
    const float out[4] = {acc0, acc1, acc2, acc3};
    if (a.slab) {
        float* __restrict__ sl = a.slab + (int64_t)blockIdx.y * a.OCp * a.Mp;
#pragma unroll
        for (int c = 0; c < 4; ++c) sl[(int64_t)c * a.Mp + m] = out[c];   // Mp is a multiple of 64
        return;
    }
    if (m >= a.M) return;
    float* __restrict__ yb = a.y + a.y_off + (int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)oh * a.y_sh + (int64_t)ow * a.y_sw;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < a.OC) {
            float v = out[c];
            float* q = yb + (int64_t)c * a.y_sc;
            if (a.accumulate) v += *q;
            *q = apply_act(v, a.act, a.slope);
        }
}

// Structured thin gather: all T taps of a channel are walked with per-tap voffsets fixed for the
// whole loop (padding folded in), the channel advances by a scalar soffset, the weights of one
// channel (T x 4 floats, K-major packed) are scalar loads.  Per (channel, tap): 1 buffer load +
// NOC FMAs, no index arithmetic.  64 positions per block; the 4 waves split the channels.
template <int T, int NOC>
__global__ __launch_bounds__(256) void thin_struct_kernel(const GatherArgs a, int RC, int rc_per_split) {
    __shared__ float red[3][4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 64;
    const uint32_t n0 = fdiv((uint32_t)m0, a.div_sp);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (int64_t)n0 * a.x_sn), 0, 0x80000000u, 0x00020000);
    const int m = m0 + lane;
    int xbase4 = 0;
    uint32_t vmask = 0;
    uint32_t n = 0, od = 0, oh = 0, ow = 0;
    if (m < a.M) {
        n = fdiv((uint32_t)m, a.div_sp);
        uint32_t r = (uint32_t)m - n * a.div_sp.div;
        od = fdiv(r, a.div_hw);
        r -= od * a.div_hw.div;
        oh = fdiv(r, a.div_w);
        ow = r - oh * a.div_w.div;
        vmask = dim_mask(a.td, (int)od, 0) | dim_mask(a.th, (int)oh, 8) | dim_mask(a.tw, (int)ow, 16);
        xbase4 = 4 * ((int)((int64_t)(n - n0) * a.x_sn) + ((int)od * a.td.mul + a.td.base) * a.x_sd +
                      ((int)oh * a.th.mul + a.th.base) * a.x_sh + ((int)ow * a.tw.mul + a.tw.base) * a.x_sw);
    }
    uint32_t vloc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t sel = a.s_sel[t];
        vloc[t] = ((vmask & sel) == sel) ? (uint32_t)(xbase4 + a.s_local[t]) : 0x80000000u;
    }
    // channels of this block's split, then of this wave
    const int cb0 = a.slab ? blockIdx.y * rc_per_split : 0;
    const int cb1 = a.slab ? min(RC, cb0 + rc_per_split) : RC;
    const int cper = (cb1 - cb0 + 3) / 4;
    const int c0 = cb0 + wave * cper, c1 = min(cb1, c0 + cper);
    float acc[NOC];
#pragma unroll
    for (int c = 0; c < NOC; ++c) acc[c] = 0.f;
    for (int rc = c0; rc < c1; ++rc) {
        const int soff = rc * a.s_stepA;
        const float4* __restrict__ wrow = reinterpret_cast<const float4*>(a.wp) + (int64_t)rc * T;   // wave-uniform
        float xv[T];
#pragma unroll
        for (int t = 0; t < T; ++t) xv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vloc[t], soff, 0));
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float4 w = wrow[t];
            acc[0] += w.x * xv[t];
            if (NOC > 1) acc[1] += w.y * xv[t];
            if (NOC > 2) acc[2] += w.z * xv[t];
            if (NOC > 3) acc[3] += w.w * xv[t];
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int c = 0; c < NOC; ++c) red[wave - 1][c][lane] = acc[c];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int c = 0; c < NOC; ++c) acc[c] += red[w][c][lane];
    if (a.slab) {
        float* __restrict__ sl = a.slab + (int64_t)blockIdx.y * a.OCp * a.Mp;
#pragma unroll
        for (int c = 0; c < 4; ++c) sl[(int64_t)c * a.Mp + m] = c < NOC ? acc[c < NOC ? c : 0] : 0.f;
        return;
    }
    if (m >= a.M) return;
    float* __restrict__ yb = a.y + a.y_off + (int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)oh * a.y_sh + (int64_t)ow * a.y_sw;
#pragma unroll
    for (int c = 0; c < NOC; ++c)
        if (c < a.OC) {
            float v = acc[c];
            float* q = yb + (int64_t)c * a.y_sc;
            if (a.accumulate) v += *q;
            *q = apply_act(v, a.act, a.slope);
        }
}

// Row-block kernels (thin_rows / thin_quad): workgroup ids go round-robin over the 8 XCDs, so with the plain plane-major numbering
// the row blocks of one plane — which share their halo rows — land on 8 different L2s and every halo row is fetched from HBM
// twice (the RGB head read 1.47x its input).  This map hands whole planes to an XCD: id b = 8 k + x runs (plane x + 8 (k / per),
// row block k % per); the planes past the last full group of 8 keep the plain numbering.
__device__ __forceinline__ void xcd_plane_map(uint32_t b, uint32_t per, uint32_t nblocks, uint32_t& plane, uint32_t& rb) {
    const uint32_t full = nblocks / (8 * per) * (8 * per);
    if (b < full) {
        const uint32_t x = b & 7, k = b >> 3;
        const uint32_t g = k / per;
        plane = x + 8 * g;
        rb = k - g * per;
    } else {
        plane = b / per;
        rb = b - plane * per;
    }
}

// --------------------------------------------------------------------------- //
// thin_rows_kernel: OC <= 4, unit stride, rows one wave (64) or half a wave (32) wide, 2-3 taps per spatial
// dim, 1 or 4 depth taps — the colour generator's RGB head (128 -> 3, 3x3 at 64x64), the data gradient of its
// 1 -> 64 stem, the depth head of the geometry generator and the data gradients of the 3-D discriminators'
// stems (stride-parity classes of 4x4(x4) filters: 2x2(x4) taps on 32-wide class grids).  thin_struct_kernel
// is L1-bound there (one dword gather per tap, channel and position).  Here a lane owns one column and FOUR
// consecutive output rows: per channel (and depth tap) the NH + 3 input rows they share are loaded once
// (coalesced rows), the left / right neighbours come from DPP wave shifts (the wave's end lanes receive the
// zero padding through bound_ctrl; 32-wide rows mask the lanes at the seam), and 4 x NH x NW x NOC FMAs run
// on them.  Depth taps that fall outside the clip are skipped per workgroup.  The 4 waves split the channels.
// --------------------------------------------------------------------------- //
template <int NOC, int NH, int NW, int ND, bool W32, int IW0, bool BN = false>
__global__ __launch_bounds__(256) void thin_rows_kernel(const GatherArgs a, int RC, const BnView bv = BnView()) {
    constexpr int NR = NH + 3, T = ND * NH * NW, RPB = W32 ? 8 : 4;   // input rows per lane, taps per channel, output rows per block
    __shared__ float red[3][4 * NOC][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = W32 ? (lane & 31) : lane;
    const uint32_t ohq = (uint32_t)a.OH / RPB;
    uint32_t plane, rblk;
    xcd_plane_map(blockIdx.x, ohq, gridDim.x, plane, rblk);
    const int oh0 = (int)rblk * RPB + (W32 ? 4 * (lane >> 5) : 0);   // this lane's first output row
    const uint32_t n = plane / (uint32_t)a.OD, od = plane - n * (uint32_t)a.OD;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)n * a.x_sn), 0, 0x80000000u, 0x00020000);
    uint32_t vrow[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int ih = oh0 + a.p_ihmin + q;
        vrow[q] = (unsigned)ih < (unsigned)a.th.size ? (uint32_t)(4 * (ih * a.x_sh + col)) : 0x80000000u;
    }
    // depth taps: byte offset of the gathered depth slice, or -1 when it is outside the clip (block-uniform)
    int doff[ND];
#pragma unroll
    for (int ud = 0; ud < ND; ++ud) {
        const int id = (int)od * a.td.mul + a.td.base + a.td.delta[ud];
        doff[ud] = (unsigned)id < (unsigned)a.td.size ? 4 * id * a.x_sd : -1;
    }
    int perm[T];   // packed-weight tap of (depth tap, row offset, column offset)
    {
        typedef int32_t i32x16 __attribute__((ext_vector_type(16)));
        const i32x16 sl16 = *reinterpret_cast<const i32x16*>(a.s_local);
#pragma unroll
        for (int t = 0; t < T; ++t) perm[t] = sl16[t];
    }
    float acc[4][NOC];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < NOC; ++c) acc[p][c] = 0.f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // BN: channels below bv.cbn are read from the BatchNorm's INPUT and normalised + activated here (their slice of the operand was never written);
    // rows outside the image must stay the convolution's zero padding, so the transform applies to rows that exist only
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t bxrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BN ? bv.bx + (int64_t)n * bv.bx_sn : a.x), 0, 0x80000000u, 0x00020000);
    [[maybe_unused]] uint32_t rowok = 0;
    if constexpr (BN) {
#pragma unroll
        for (int q = 0; q < NR; ++q) rowok |= (vrow[q] != 0x80000000u) ? 1u << q : 0u;      // (row validity is the same for every lane of a 64-wide row)
    }
    for (int rc = wave; rc < RC; rc += 4) {
        const f32x4* __restrict__ wrow = reinterpret_cast<const f32x4*>(a.wp) + (int64_t)rc * T;   // wave-uniform
        [[maybe_unused]] const bool bnch = BN && rc < bv.cbn;
        [[maybe_unused]] float bsc = 1.f, bsh = 0.f;
        if constexpr (BN) {
            if (bnch) {
                const float is = bv.invstd[rc];
                bsc = bv.gamma[rc] * is;
                bsh = bv.beta[rc] - bv.mean[rc] * bsc;      // BnApply's expressions, operation for operation
            }
        }
#pragma unroll
        for (int ud = 0; ud < ND; ++ud) {
            if (doff[ud] < 0) continue;
            const int soff = rc * a.s_stepA + doff[ud];
            float r[NR];
            if (BN && bnch) {
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bxrs, vrow[q], rc * (bv.bx_sc * 4), 0));
                    const float z = v * bsc + bsh;
                    const float t = bv.act == DCV_ACT_LEAKY ? (z > 0.f ? z : z * bv.slope) : z;
                    r[q] = ((rowok >> q) & 1u) ? t : 0.f;
                }
            } else {
#pragma unroll
            for (int q = 0; q < NR; ++q) r[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vrow[q], soff, 0));
            }
            f32x4 w[NH * NW];
#pragma unroll
            for (int t = 0; t < NH * NW; ++t) w[t] = wrow[perm[ud * NH * NW + t]];
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                // columns col + IW0 + b, b = 0 .. NW-1, as offsets -1 / 0 / +1 from this lane's column
                float v[3];
                v[1] = r[q];
                v[0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r[q]), 0x138, 0xf, 0xf, true));   // wave_shr:1
                v[2] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r[q]), 0x130, 0xf, 0xf, true));   // wave_shl:1
                if (W32) {   // the seam between the two 32-wide rows of the wave is padding, not a neighbour
                    v[0] = col == 0 ? 0.f : v[0];
                    v[2] = col == 31 ? 0.f : v[2];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int ra = q - p;   // row offset of input row q for output row p
                    if (ra < 0 || ra >= NH) continue;
#pragma unroll
                    for (int b = 0; b < NW; ++b) {
                        const f32x4 wt = w[ra * NW + b];
                        const float xv = v[IW0 + b + 1];
                        acc[p][0] += wt[0] * xv;
                        if (NOC > 1) acc[p][1] += wt[1] * xv;
                        if (NOC > 2) acc[p][2] += wt[2] * xv;
                        if (NOC > 3) acc[p][3] += wt[3] * xv;
                    }
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < NOC; ++c) red[wave - 1][p * NOC + c][lane] = acc[p][c];
    }
    __syncthreads();
    if (wave != 0) return;
    const int OC = a.OC, act = a.act, accumulate = a.accumulate;
    const float slope = a.slope;
    float* __restrict__ yb = a.y + a.y_off + (int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)col * a.y_sw;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < NOC; ++c) {
            if (c >= OC) continue;
            float v = acc[p][c] + ((red[0][p * NOC + c][lane] + red[1][p * NOC + c][lane]) + red[2][p * NOC + c][lane]);
            float* q = yb + (int64_t)(oh0 + p) * a.y_sh + (int64_t)c * a.y_sc;
            if (accumulate) v += *q;
            *q = apply_act(v, act, slope);
        }
}

template <int NH, int NW, int ND, bool W32, int IW0>
static void launch_thin_rows(const GatherArgs& a, int OC, int RC, dim3 grid, hipStream_t s) {
    switch (OC) {
        case 1: hipLaunchKernelGGL((thin_rows_kernel<1, NH, NW, ND, W32, IW0>), grid, dim3(256), 0, s, a, RC); break;
        case 2: hipLaunchKernelGGL((thin_rows_kernel<2, NH, NW, ND, W32, IW0>), grid, dim3(256), 0, s, a, RC); break;
        case 3: hipLaunchKernelGGL((thin_rows_kernel<3, NH, NW, ND, W32, IW0>), grid, dim3(256), 0, s, a, RC); break;
        default: hipLaunchKernelGGL((thin_rows_kernel<4, NH, NW, ND, W32, IW0>), grid, dim3(256), 0, s, a, RC); break;
    }
}

// --------------------------------------------------------------------------- //
// widen_rows_kernel: the mirror case — at most 4 GATHERED channels, many output channels, 3x3 taps, unit stride,
// 64-wide rows: the data gradient of the colour generator's RGB head (3 -> 128 at 64x64; 27 multiply-adds per
// output, the MFMA path pads K = 27 to 32 and runs at the store rate of a 2.35 GB tensor).  A lane owns one
// column and 4 output rows; the RC x 6 input rows with their DPP neighbours are read ONCE per workgroup into
// registers, the packed weights of the whole op (27 x OC) into LDS, and the 4 waves walk the output channels:
// 108 FMAs and 4 coalesced row stores per channel, no operand traffic in the loop.
// --------------------------------------------------------------------------- //
template <int RC>
__global__ __launch_bounds__(256) void widen_rows_kernel(const GatherArgs a) {
    extern __shared__ float wl[];   // [RC * 9][OCp] packed weights
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ohq = (uint32_t)a.OH >> 2;
    const uint32_t plane = blockIdx.x / ohq;
    const int oh0 = (int)(blockIdx.x - plane * ohq) * 4;
    const uint32_t n = plane / (uint32_t)a.OD, od = plane - n * (uint32_t)a.OD;
    for (int e = tid; e < RC * 9 * a.OCp; e += 256) wl[e] = a.wp[e];
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (int64_t)n * a.x_sn + (int64_t)((int)od * a.td.mul + a.td.base + a.td.delta[0]) * a.x_sd), 0, 0x80000000u, 0x00020000);
    float v[RC][6][3];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int ih = oh0 + a.p_ihmin + q;
        const uint32_t vo = (unsigned)ih < (unsigned)a.th.size ? (uint32_t)(4 * (ih * a.x_sh + lane)) : 0x80000000u;
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
            const float c = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vo, rc * a.s_stepA, 0));
            v[rc][q][1] = c;
            v[rc][q][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x138, 0xf, 0xf, true));   // column - 1
            v[rc][q][2] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x130, 0xf, 0xf, true));   // column + 1
        }
    }
    int perm[9];
    {
        typedef int32_t i32x16 __attribute__((ext_vector_type(16)));
        const i32x16 sl16 = *reinterpret_cast<const i32x16*>(a.s_local);
#pragma unroll
        for (int t = 0; t < 9; ++t) perm[t] = sl16[t];
    }
    __syncthreads();
    const int OC = a.OC, OCp = a.OCp, act = a.act, accumulate = a.accumulate;
    const float slope = a.slope;
    float* __restrict__ yb = a.y + a.y_off + (int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)lane * a.y_sw;
    if constexpr (RC >= 3) {
    for (int oc = wave; oc < OC; oc += 4) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rc = 0; rc < RC; ++rc)
#pragma unroll
            for (int ra = 0; ra < 3; ++ra)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float w = wl[(rc * 9 + perm[ra * 3 + b]) * OCp + oc];   // wave-uniform address: LDS broadcast
#pragma unroll
                    for (int p = 0; p < 4; ++p) o[p] += w * v[rc][p + ra][b];
                }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float* q = yb + (int64_t)(oh0 + p) * a.y_sh + (int64_t)oc * a.y_sc;
            float x = o[p];
            if (accumulate) x += *q;
            *q = apply_act(x, act, slope);
        }
    }
    return;
    }
    // 1 - 2 gathered channels: two output channels at a time (OCp is even): one 8-byte LDS broadcast per tap and packed multiply-adds
    // (v_pk_fma_f32 with the input value broadcast to both halves); 0.230 -> 0.201 ms on the colour stem.  With 3 gathered channels
    // (the RGB head's data gradient) the paired loop measured slower (0.71 -> 0.83 ms) and the one-channel loop above stays.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (int oc = 2 * wave; oc < OC; oc += 8) {
        f32x2 o[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) o[p] = f32x2{0.f, 0.f};
#pragma unroll
        for (int rc = 0; rc < RC; ++rc)
#pragma unroll
            for (int ra = 0; ra < 3; ++ra)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const f32x2 w = *reinterpret_cast<const f32x2*>(&wl[(rc * 9 + perm[ra * 3 + b]) * OCp + oc]);   // wave-uniform address: LDS broadcast
#pragma unroll
                    for (int p = 0; p < 4; ++p) o[p] += w * v[rc][p + ra][b];
                }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (oc + h >= OC) break;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float* q = yb + (int64_t)(oh0 + p) * a.y_sh + (int64_t)(oc + h) * a.y_sc;
                float x = o[p][h];
                if (accumulate) x += *q;
                *q = apply_act(x, act, slope);
            }
        }
    }
}

// --------------------------------------------------------------------------- //
// widen_mfma_kernel (round 6): the same op on the matrix pipe.  widen_rows_kernel<3> spends 27 VALU multiply-adds plus a broadcast LDS read per 4 of
// them on every output element and ran the RGB head's data gradient at 3.5 TB/s of stores (0.68 ms for 2.35 GB).  As a GEMM the op is
// [OC x K] x [K x positions] with K = 9 RC (27, padded to 28): 14 v_mfma_f32_32x32x2_f32 per 32 channels x 32 positions, a quarter of the store time at the
// HBM rate, so the kernel is bound by its stores.  One wave owns one output row: the A fragments (the packed weights of ALL 32 MT channels: 14 MT registers)
// are staged in LDS once per workgroup; per half row the wave loads its 14 B values straight from the gathered tensor (lane = column, lane half = k parity; rows
// outside the image and the two halo columns arrive as zeros through the out-of-range buffer offset; the tensor is 3 channels: every load is an L1/L2 hit),
// runs 14 MFMAs per channel tile and stores its 16 registers, each one 128-byte run of a channel row per lane half.  k order = (rc, row tap, column tap) ascending, the
// order widen_rows_kernel's fma chain uses: results are bit-identical to it.
// --------------------------------------------------------------------------- //
template <int RC, int MT>
__global__ __launch_bounds__(256, 4) void widen_mfma_kernel(const GatherArgs a, int groups) {
    constexpr int K = RC * 9, KS = (K + 1) / 2;
    __shared__ float wl[2 * KS][MT * 32];              // weights in natural k order (rc, row tap, column tap); the padding row is zero
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int e = tid; e < 2 * KS * MT * 32; e += 256) {
        const int k = e / (MT * 32), oc = e - k * (MT * 32);
        const int rc = k / 9, t = k - rc * 9;
        int pt = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) pt = (t == q) ? a.s_local[q] : pt;
        wl[k][oc] = (k < K && oc < a.OCp) ? a.wp[(int64_t)(rc * 9 + pt) * a.OCp + oc] : 0.f;
    }
    const uint32_t gpp = ((uint32_t)a.OH >> 2) / (uint32_t)groups;      // workgroups per plane
    uint32_t plane, gb;
    xcd_plane_map(blockIdx.x, gpp, gridDim.x, plane, gb);
    const uint32_t n = plane / (uint32_t)a.OD, od = plane - n * (uint32_t)a.OD;
    // this lane's k of MFMA step m is 2 m + lhi
    int32_t kb4[KS];        // byte offset of that k inside the gathered tensor, relative to (first tap row, column - 1)
    // per MFMA step (bit m): the lane's k is padding / has column tap 0 / column tap 2 / row tap 0, 1, 2
    uint32_t padmask = 0, b0mask = 0, b2mask = 0, r0mask = 0, r1mask = 0, r2mask = 0;
#pragma unroll
    for (int m = 0; m < KS; ++m) {
        const int k = 2 * m + lhi;
        const bool live = k < K;
        const int kk = live ? k : 0;
        const int rc = kk / 9, t = kk - rc * 9, ra = t / 3, b = t - ra * 3;
        kb4[m] = rc * a.s_stepA + 4 * (ra * a.x_sh + b);
        padmask |= live ? 0u : 1u << m;
        b0mask |= (live && b == 0) ? 1u << m : 0u;
        b2mask |= (live && b == 2) ? 1u << m : 0u;
        r0mask |= (live && ra == 0) ? 1u << m : 0u;
        r1mask |= (live && ra == 1) ? 1u << m : 0u;
        r2mask |= (live && ra == 2) ? 1u << m : 0u;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (int64_t)n * a.x_sn + (int64_t)((int)od * a.td.mul + a.td.base + a.td.delta[0]) * a.x_sd), 0, 0x80000000u, 0x00020000);
    const int OC = a.OC, act = a.act, accumulate = a.accumulate;
    const float slope = a.slope;
    // (run_gather has checked that a sample's output spans < 2^29 elements and OCp channel strides < 2^30)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y + a.y_off + (int64_t)n * a.y_sn + (int64_t)od * a.y_sd, 0, 0x80000000u, 0x00020000);
    const int ysh = (int)a.y_sh, ysw = (int)a.y_sw;
    const uint32_t ysc4 = (uint32_t)a.y_sc * 4u;
    const bool full = OC == MT * 32;

    auto load_b = [&](int oh, int x0, float (&bv)[KS]) {
        // rows oh + p_ihmin + ra (validity is wave-uniform), columns x0 + l31 + b - 1 (only the image's first and last column have a halo tap)
        const int ih0 = oh + a.p_ihmin;
        const int col = x0 + l31;
        uint32_t bad = padmask | (col == 0 ? b0mask : 0u) | (col == a.tw.size - 1 ? b2mask : 0u);
        if ((unsigned)ih0 >= (unsigned)a.th.size) bad |= r0mask;
        if ((unsigned)(ih0 + 1) >= (unsigned)a.th.size) bad |= r1mask;
        if ((unsigned)(ih0 + 2) >= (unsigned)a.th.size) bad |= r2mask;
        const int base4 = 4 * (ih0 * a.x_sh + col - 1);
#pragma unroll
        for (int m = 0; m < KS; ++m) {
            const uint32_t vo = ((bad >> m) & 1u) ? 0x80000000u : (uint32_t)(base4 + kb4[m]);
            bv[m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vo, 0, 0));
        }
    };

    const int ntile = groups * 2;              // (row, half row) pairs of this wave
    __syncthreads();
    // Channel tile outermost, the wave's `groups` CONSECUTIVE rows inside: a channel-row run of groups x 256 bytes is written back to back by one wave and the
    // four waves' runs adjoin, instead of 128 bytes per channel and tile (measured, B = 70: 0.66 -> 0.57 ms); non-temporal stores (nothing reads the tile
    // back through this L2): 0.57 -> 0.53 ms.  B is re-read per channel tile (L1 hits: the workgroup's 3 x 18 gathered rows are 14 KB).
    const int ohw = (int)gb * groups * 4 + wave * groups;
#pragma unroll 1
    for (int mt = 0; mt < MT; ++mt) {
        float af[KS];
#pragma unroll
        for (int m = 0; m < KS; ++m) af[m] = wl[2 * m + lhi][mt * 32 + l31];
        float bcur[KS], bnxt[KS];
        load_b(ohw, 0, bcur);
#pragma unroll 1
        for (int tt = 0; tt < ntile; ++tt) {
            const int oh = ohw + (tt >> 1), x0 = (tt & 1) * 32;
            if (tt + 1 < ntile) load_b(ohw + ((tt + 1) >> 1), ((tt + 1) & 1) * 32, bnxt);
            // stores: one 32-bit offset per lane and tile, the channel of (mt, r) as the scalar offset (no per-register 64-bit addresses to keep alive)
            const uint32_t yv = (uint32_t)(4 * (oh * ysh + (x0 + l31) * ysw)) + (uint32_t)(4 * lhi) * ysc4;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int m = 0; m < KS; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bcur[m], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ocb = mt * 32 + (r & 3) + 8 * (r >> 2);
                const uint32_t vo = (full || ocb + 4 * lhi < OC) ? yv : 0x80000000u;
                const uint32_t so = (uint32_t)ocb * ysc4;
                float v = acc[r];
                if (accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, vo, so, 0));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, apply_act(v, act, slope)), yrs, vo, so, 2);   // aux 2 = nt
            }
#pragma unroll
            for (int m = 0; m < KS; ++m) bcur[m] = bnxt[m];
        }
    }
}

// --------------------------------------------------------------------------- //
// head_bn_kernel (round 6): widen_mfma_kernel<3, 4> fused with the backward of the BatchNorm + (Leaky)ReLU that produced the head's first `cbn` input channels —
// the last stage of the colour generator: UpBlock 5's BatchNorm -> concat with the stem's skip -> Outconv (generator.py:238-282,393-400).  Unfused, the head's data
// gradient WRITES 2.35 GB, BatchNorm's reduction reads half of it back with the BatchNorm input (2.35 GB) and its apply pass reads both again and writes 1.17 GB:
// 8.2 GB per backward.  A tile of that gradient is 14 MFMAs away from the 3-channel cotangent, so the first half never has to exist in memory:
//   MODE 1  recomputes every tile; for the BatchNorm channels it loads the matching tile of the BatchNorm INPUT (same register layout as the stores: 128-byte
//           runs of a channel row), forms dz = dx * act'(z) with z = x * sc + sh exactly as the forward did, and accumulates sum dz and sum dz (x - mean) per
//           channel (lanes by DPP, waves through LDS, one partial row per workgroup); the other channels' tiles are stored as before;
//   MODE 2  (after the finalize) recomputes the BatchNorm channels' tiles once more and writes the gradient of the BatchNorm INPUT directly:
//           sc dz + c0 + c1 (x - mean), c0 = -sc sum(dz) / n, c1 = -sc invstd^2 sum(dz (x - mean)) / n.
// 4.7 GB instead of 8.2 GB, two launches + a finalize instead of widen + reduce + finalize + apply.
// --------------------------------------------------------------------------- //
struct HeadBnArgs {
    GatherArgs g;
    const float* bx;                 // BatchNorm input (cbn channels, the head's spatial size)
    float* bdx;                      // MODE 2: its gradient
    const float* gamma; const float* beta; const float* mean; const float* invstd;
    const float* cst;                // MODE 2: [cbn][2] = {c0, c1}
    float* partial;                  // MODE 1: [workgroups][cbn][2]
    int64_t bx_sn, bdx_sn;
    int32_t bx_sc, bx_sh, bx_sw, bdx_sc, bdx_sh, bdx_sw;
    int32_t cbn, groups;
    float slope; int32_t pad;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void head_bn_kernel(const HeadBnArgs h) {
    constexpr int RC = 3, MT = 4, K = RC * 9, KS = (K + 1) / 2;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ float wl[2 * KS][MT * 32];
    __shared__ f32x4 c4[MT * 32];            // per BatchNorm channel: sc = gamma * invstd, sh = beta - mean * sc, mean, (unused)
    __shared__ f32x2 c2[MT * 32];            // MODE 2: c0, c1
    __shared__ float red[4][MT * 32][2];     // MODE 1: per-wave channel sums
    const GatherArgs& a = h.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int cbn = h.cbn, MB = cbn >> 5;      // BatchNorm channel tiles
    for (int e = tid; e < 2 * KS * MT * 32; e += 256) {
        const int k = e / (MT * 32), oc = e - k * (MT * 32);
        const int rc = k / 9, t = k - rc * 9;
        int pt = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) pt = (t == q) ? a.s_local[q] : pt;
        wl[k][oc] = (k < K && oc < a.OCp) ? a.wp[(int64_t)(rc * 9 + pt) * a.OCp + oc] : 0.f;
    }
    if (tid < cbn) {
        const float is = h.invstd[tid], mu = h.mean[tid];
        const float sc = h.gamma[tid] * is, sh = h.beta[tid] - mu * sc;      // BnApply's own expressions: the branch of the activation must be the forward's
        c4[tid] = f32x4{sc, sh, mu, 0.f};
        if (MODE == 2) c2[tid] = f32x2{h.cst[2 * tid], h.cst[2 * tid + 1]};
    }
    if (MODE == 1)
        for (int e = tid; e < 4 * MT * 32 * 2; e += 256) (&red[0][0][0])[e] = 0.f;
    const int groups = h.groups;
    const uint32_t gpp = ((uint32_t)a.OH >> 2) / (uint32_t)groups;
    uint32_t plane, gb;
    xcd_plane_map(blockIdx.x, gpp, gridDim.x, plane, gb);
    const uint32_t n = plane / (uint32_t)a.OD, od = plane - n * (uint32_t)a.OD;
    int32_t kb4[KS];
    uint32_t padmask = 0, b0mask = 0, b2mask = 0, r0mask = 0, r1mask = 0, r2mask = 0;
#pragma unroll
    for (int m = 0; m < KS; ++m) {
        const int k = 2 * m + lhi;
        const bool live = k < K;
        const int kk = live ? k : 0;
        const int rc = kk / 9, t = kk - rc * 9, ra = t / 3, b = t - ra * 3;
        kb4[m] = rc * a.s_stepA + 4 * (ra * a.x_sh + b);
        padmask |= live ? 0u : 1u << m;
        b0mask |= (live && b == 0) ? 1u << m : 0u;
        b2mask |= (live && b == 2) ? 1u << m : 0u;
        r0mask |= (live && ra == 0) ? 1u << m : 0u;
        r1mask |= (live && ra == 1) ? 1u << m : 0u;
        r2mask |= (live && ra == 2) ? 1u << m : 0u;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (int64_t)n * a.x_sn + (int64_t)((int)od * a.td.mul + a.td.base + a.td.delta[0]) * a.x_sd), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y + a.y_off + (int64_t)n * a.y_sn + (int64_t)od * a.y_sd, 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t bxrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(h.bx + (int64_t)n * h.bx_sn), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t bdrs = __builtin_amdgcn_make_buffer_rsrc((MODE == 2 ? h.bdx : a.y) + (MODE == 2 ? (int64_t)n * h.bdx_sn : 0), 0, 0x80000000u, 0x00020000);
    const int ysh = (int)a.y_sh, ysw = (int)a.y_sw;
    const uint32_t ysc4 = (uint32_t)a.y_sc * 4u, bxsc4 = (uint32_t)h.bx_sc * 4u, bdsc4 = (uint32_t)h.bdx_sc * 4u;
    const float slope = h.slope;

    auto load_b = [&](int oh, int x0, float (&bv)[KS]) {
        const int ih0 = oh + a.p_ihmin;
        const int col = x0 + l31;
        uint32_t bad = padmask | (col == 0 ? b0mask : 0u) | (col == a.tw.size - 1 ? b2mask : 0u);
        if ((unsigned)ih0 >= (unsigned)a.th.size) bad |= r0mask;
        if ((unsigned)(ih0 + 1) >= (unsigned)a.th.size) bad |= r1mask;
        if ((unsigned)(ih0 + 2) >= (unsigned)a.th.size) bad |= r2mask;
        const int base4 = 4 * (ih0 * a.x_sh + col - 1);
#pragma unroll
        for (int m = 0; m < KS; ++m) {
            const uint32_t vo = ((bad >> m) & 1u) ? 0x80000000u : (uint32_t)(base4 + kb4[m]);
            bv[m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vo, 0, 0));
        }
    };

    const int ntile = groups * 2;
    __syncthreads();
    const int ohw = (int)gb * groups * 4 + wave * groups;
    const int mt_end = MODE == 2 ? MB : MT;
#pragma unroll 1
    for (int mt = 0; mt < mt_end; ++mt) {
        const bool bn = mt < MB;               // wave-uniform
        float af[KS];
#pragma unroll
        for (int m = 0; m < KS; ++m) af[m] = wl[2 * m + lhi][mt * 32 + l31];
        float s1[16], s2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
        float bcur[KS], bnxt[KS];
        load_b(ohw, 0, bcur);
#pragma unroll 1
        for (int tt = 0; tt < ntile; ++tt) {
            const int oh = ohw + (tt >> 1), x0 = (tt & 1) * 32;
            // the BatchNorm input's tile, in the accumulators' layout, issued ahead of the MFMAs that hide its latency
            float xt[16];
            if (bn) {
                const uint32_t xv = (uint32_t)(4 * (oh * h.bx_sh + (x0 + l31) * h.bx_sw)) + (uint32_t)(4 * lhi) * bxsc4;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    xt[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bxrs, xv, (uint32_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * bxsc4, 0));
            }
            if (tt + 1 < ntile) load_b(ohw + ((tt + 1) >> 1), ((tt + 1) & 1) * 32, bnxt);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int m = 0; m < KS; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bcur[m], acc, 0, 0, 0);
            if (bn) {
                const uint32_t dv = (uint32_t)(4 * (oh * h.bdx_sh + (x0 + l31) * h.bdx_sw)) + (uint32_t)(4 * lhi) * bdsc4;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const f32x4 k4 = c4[ch];
                    const float z = xt[r] * k4[0] + k4[1];
                    const float dz = acc[r] * (z > 0.f ? 1.f : slope);
                    const float xm = xt[r] - k4[2];
                    if (MODE == 1) {
                        s1[r] += dz;
                        s2[r] += dz * xm;
                    } else {
                        const f32x2 k2 = c2[ch];
                        const float o = k4[0] * dz + k2[0] + k2[1] * xm;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, o), bdrs, dv, (uint32_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * bdsc4, 2);
                    }
                }
            } else {
                const uint32_t yv = (uint32_t)(4 * (oh * ysh + (x0 + l31) * ysw)) + (uint32_t)(4 * lhi) * ysc4;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[r];      // (a copy: __builtin_bit_cast applied to the vector ELEMENT expression itself compiled to element 0 for every r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), yrs, yv, (uint32_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * ysc4, 2);
                }
            }
#pragma unroll
            for (int m = 0; m < KS; ++m) bcur[m] = bnxt[m];
        }
        if (MODE == 1 && bn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t1 = half_wave_sum(s1[r]), t2 = half_wave_sum(s2[r]);
                if (l31 == 31) {
                    const int ch = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    red[wave][ch][0] = t1;
                    red[wave][ch][1] = t2;
                }
            }
        }
    }
    if (MODE == 1) {
        __syncthreads();
        if (tid < 2 * cbn) {
            const int ch = tid >> 1, w = tid & 1;
            h.partial[((int64_t)blockIdx.x * cbn + ch) * 2 + w] = ((red[0][ch][w] + red[1][ch][w]) + red[2][ch][w]) + red[3][ch][w];
        }
    }
}

// partial[workgroups][cbn][2] -> dbeta = sum dz, dgamma = invstd sum dz (x - mean); cst[cbn][2] for head_bn_kernel<2>.  One workgroup per channel, fp64, fixed order.
__global__ __launch_bounds__(256) void head_bn_finalize_kernel(const float* __restrict__ partial, int nwg, int cbn, double count, const float* __restrict__ gamma,
                                                               const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ cst) {
    __shared__ double red[2][256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double a0 = 0.0, a1 = 0.0;
    for (int w = tid; w < nwg; w += 256) {
        a0 += (double)partial[((int64_t)w * cbn + c) * 2];
        a1 += (double)partial[((int64_t)w * cbn + c) * 2 + 1];
    }
    red[0][tid] = a0; red[1][tid] = a1;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double is = (double)invstd[c], sc = (double)gamma[c] * is;
        const double S1 = red[0][0], S2 = red[1][0];
        dbeta[c] = (float)S1;
        dgamma[c] = (float)(S2 * is);
        cst[2 * c] = (float)(-sc * S1 / count);
        cst[2 * c + 1] = (float)(-sc * is * is * S2 / count);
    }
}

// --------------------------------------------------------------------------- //
// thin_quad_kernel: OC <= 4 outputs of a 4x4, stride-2, pad-1 SCATTER-form op (conv data gradient / transposed-conv forward)
// on 32-wide gathered rows -> 64-wide output rows, 1 or 4 depth taps (stride 1, no depth padding): the geometry generator's
// depth / flow head and the data gradients of the 3-D discriminators' stems.  thin_rows_kernel runs these as four stride-parity
// classes, each of which re-reads the same gathered rows (20 row loads per 64 multiply-adds and channel); here a lane owns the
// 2x2 output quads of one gathered column for two quad rows: the 4 gathered rows it needs are loaded once per (channel, depth
// tap), their left / right neighbours come from DPP wave shifts, and all 16 taps are applied — 4 row loads per 32 multiply-adds,
// and the two outputs of a quad row are stored as one 8-byte pair (whole 256-byte row segments per half-wave).
// Block = 4 waves x (2 half-waves x 2 quad rows) = 8 output rows of one (n, output depth) plane; the waves split the channels.
// --------------------------------------------------------------------------- //
struct QuadArgs {
    const float* s;
    float* y;
    const float* w;
    int32_t N, RC, OC, SD, SH, OD, pad0, pad1;
    int64_t s_sn, s_sc, s_sd, s_sh;
    int64_t y_sn, y_sc, y_sd, y_sh;
    int64_t w_o, w_r;      // weight element (output channel c, gathered channel rc, kd, kh, kw) at c * w_o + rc * w_r + (kd * 4 + kh) * 4 + kw
    int32_t act, accumulate;
    float slope, pad2;
};

template <int NOC, int ND>
__global__ __launch_bounds__(256) void thin_quad_kernel(const QuadArgs a) {
    __shared__ float red[3][8 * NOC][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hw = lane >> 5;
    const uint32_t qb = (uint32_t)a.SH >> 2;                       // blocks per plane: 4 quad rows each
    const uint32_t plane = blockIdx.x / qb;                        // (plain numbering: the XCD plane map of thin_rows_kernel measured slower here,
    const int r0 = (int)(blockIdx.x - plane * qb) * 4 + 2 * hw;    //  3-D stems 0.26 -> 0.34 ms — neighbouring output depths share their input planes)
                                                                   // this half-wave's first quad row (= gathered row)
    const uint32_t n = plane / (uint32_t)a.OD, od = plane - n * (uint32_t)a.OD;
    // the 4 gathered rows r0 - 1 .. r0 + 2: element offsets inside a (channel, depth) plane, or -1 outside the tensor
    int roff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = r0 - 1 + i;
        roff[i] = (unsigned)rr < (unsigned)a.SH ? rr * (int)a.s_sh + col : -1;
    }
    float acc[2][2][2][NOC];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < NOC; ++c) acc[q][h][b][c] = 0.f;
    const float* __restrict__ sn = a.s + (int64_t)n * a.s_sn;
#pragma unroll(ND == 1 ? 4 : 1)
    for (int rc = wave; rc < a.RC; rc += 4) {
        // all row loads of this channel first (4 per depth tap), so that they are in flight together
        float ctr[ND][4];
#pragma unroll
        for (int kd = 0; kd < ND; ++kd) {
            const int sd = (int)od - kd;                           // produced depth = gathered depth + kd
            const bool in = (unsigned)sd < (unsigned)a.SD;         // block-uniform
            const float* __restrict__ pl = sn + (int64_t)rc * a.s_sc + (int64_t)(in ? sd : 0) * a.s_sd;
#pragma unroll
            for (int i = 0; i < 4; ++i) ctr[kd][i] = (in && roff[i] >= 0) ? pl[roff[i]] : 0.f;
        }
#pragma unroll
        for (int kd = 0; kd < ND; ++kd) {
            float g[4][3];                                         // [row][left, centre, right]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = ctr[kd][i];
                g[i][1] = v;
                float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));   // wave_shr:1 -> column - 1
                float r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));   // wave_shl:1 -> column + 1
                g[i][0] = col == 0 ? 0.f : l;                      // the seam between the wave's two 32-wide rows is padding
                g[i][2] = col == 31 ? 0.f : r;
            }
#pragma unroll
            for (int c = 0; c < NOC; ++c) {
                if (c >= a.OC) continue;
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4* __restrict__ wq = reinterpret_cast<const f32x4*>(a.w + (int64_t)c * a.w_o + (int64_t)rc * a.w_r + kd * 16);   // wave-uniform: w[kh][0..3]
                const f32x4 w0 = wq[0], w1 = wq[1], w2 = wq[2], w3 = wq[3];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float (&top)[3] = g[q], (&mid)[3] = g[q + 1], (&bot)[3] = g[q + 2];
                    // output row 2r     : gathered row r (kh = 1) and r - 1 (kh = 3);  output row 2r + 1: row r + 1 (kh = 0) and r (kh = 2)
                    // output col 2c     : gathered col c (kw = 1) and c - 1 (kw = 3);  output col 2c + 1: col c + 1 (kw = 0) and c (kw = 2)
                    acc[q][0][0][c] += mid[1] * w1[1] + mid[0] * w1[3] + top[1] * w3[1] + top[0] * w3[3];
                    acc[q][0][1][c] += mid[2] * w1[0] + mid[1] * w1[2] + top[2] * w3[0] + top[1] * w3[2];
                    acc[q][1][0][c] += bot[1] * w0[1] + bot[0] * w0[3] + mid[1] * w2[1] + mid[0] * w2[3];
                    acc[q][1][1][c] += bot[2] * w0[0] + bot[1] * w0[2] + mid[2] * w2[0] + mid[1] * w2[2];
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < NOC; ++c) red[wave - 1][((q * 2 + h) * 2 + b) * NOC + c][lane] = acc[q][h][b][c];
    }
    __syncthreads();
    if (wave != 0) return;
    const int OC = a.OC, act = a.act, accumulate = a.accumulate;
    const float slope = a.slope;
    float* __restrict__ yb = a.y + (int64_t)n * a.y_sn + (int64_t)od * a.y_sd + 2 * col;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < NOC; ++c) {
                if (c >= OC) continue;
                float v[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int e = ((q * 2 + h) * 2 + b) * NOC + c;
                    v[b] = acc[q][h][b][c] + ((red[0][e][lane] + red[1][e][lane]) + red[2][e][lane]);
                }
                float* dst = yb + (int64_t)(2 * (r0 + q) + h) * a.y_sh + (int64_t)c * a.y_sc;
                if (accumulate) { v[0] += dst[0]; v[1] += dst[1]; }
                dst[0] = apply_act(v[0], act, slope);
                dst[1] = apply_act(v[1], act, slope);
            }
}

template <int T>
static bool launch_thin_struct(const GatherArgs& a, int OC, int RC, int rc_per_split, dim3 grid, hipStream_t s) {
    switch (OC) {
        case 1: hipLaunchKernelGGL((thin_struct_kernel<T, 1>), grid, dim3(256), 0, s, a, RC, rc_per_split); return true;
        case 2: hipLaunchKernelGGL((thin_struct_kernel<T, 2>), grid, dim3(256), 0, s, a, RC, rc_per_split); return true;
        case 3: hipLaunchKernelGGL((thin_struct_kernel<T, 3>), grid, dim3(256), 0, s, a, RC, rc_per_split); return true;
        case 4: hipLaunchKernelGGL((thin_struct_kernel<T, 4>), grid, dim3(256), 0, s, a, RC, rc_per_split); return true;
    }
    return false;
}

// Wp[k][oc] = w[oc * ws_o + ktab[k].w_off]   (zero for padding rows / channels); blockIdx.y = class
struct PackArgs {
    const KEntry* ktab[4];
    float* wp[4];
    int32_t K16[4];
    int32_t fmt[4];   // 0: fp32 wp[k][OCp]; 1: bf16 wp16[k / 8][OCp][8] (the bf16-product LDS-DMA kernels: an MFMA A fragment is one 16-byte LDS read);
                      // 2: fp32 on the bf16 pipe: wp16[k / 16][piece 0..2][(k / 8) & 1][OCp][8], piece = hi / mid / lo of the 3-way bf16 split
};
__global__ void pack_weights_kernel(const float* __restrict__ w, const PackArgs pa, int OC, int OCp, int64_t ws_o) {
    const int c = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)pa.K16[c] * OCp) return;
    const int k = (int)(i / OCp), oc = (int)(i % OCp);
    const KEntry e = pa.ktab[c][k];
    float v = 0.f;
    if (oc < OC && !(e.tapsel >> 31)) v = w[(int64_t)oc * ws_o + e.w_off];
    if (pa.fmt[c] == 2) {   // the same RNE split as split3_bf16x8
        // K steps whose X6_PHASE bit is set carry NEGATED weights: the kernel keeps its accumulators negated while it runs them (see X6_PHASE)
        if (X6_PHASE(k >> 4)) v = -v;
        __bf16* o = reinterpret_cast<__bf16*>(pa.wp[c]);
        const __bf16 hi = (__bf16)v;
        const float r = v - (float)hi;
        const __bf16 mid = (__bf16)r;
        const __bf16 lo = (__bf16)(r - (float)mid);
        const int64_t base = (((int64_t)(k >> 4) * 6 + ((k >> 3) & 1)) * OCp + oc) * 8 + (k & 7);
        o[base] = hi; o[base + (int64_t)2 * OCp * 8] = mid; o[base + (int64_t)4 * OCp * 8] = lo;
    } else if (pa.fmt[c]) reinterpret_cast<__bf16*>(pa.wp[c])[((int64_t)(k >> 3) * OCp + oc) * 8 + (k & 7)] = (__bf16)v;   // RNE, as v_cvt_pk_bf16_f32 in the activations' fragments
    else pa.wp[c][i] = v;
}

// --------------------------------------------------------------------------- //
// wgrad GEMM.  R[dc, j] = sum_m D[dc, m] G[j, m].  Block tile BD x BJ, reduction
// step 32 positions, blockIdx.y = split of the position range.
// --------------------------------------------------------------------------- //
struct WgradArgs {
    const float* dptr;
    const float* gptr;
    float* slab;
    const KEntry* jtab;
    int32_t M, DC, J, DCp, Jp, chunk, pad0, pad1;
    FastDiv div_sp, div_hw, div_w;
    DimTaps td, th, tw;
    int64_t d_sn, g_sn;
    int32_t d_sc4, d_sd, d_sh, d_sw;   // d_sc4: channel stride in BYTES
    int32_t g_sd, g_sh, g_sw, pad2;
    // regular form (4x4 inner taps): row j = (gc * nd + kd) * 16 + t
    int32_t g_sc4, g_sd4, log2nd, pad3;
    int32_t hw_off4[16];   // byte offset of inner tap t = kh * 4 + kw
    uint32_t hw_sel[16];   // its selection bits
};

// REG16: 4x4 inner taps.  A thread's gathered rows j = j0 + sub + 8 i then use only the two inner taps
// t = sub and sub + 8 (row parity), and (channel, depth tap) = j / 16 is wave-uniform: two per-lane
// voffsets per step (padding folded in) + precomputed scalar soffsets replace 32 per-row registers,
// which is what lets three workgroups share a CU.
template <int TD, int TJ, int WD, int WJ, bool REG16>
__global__ __launch_bounds__(256, (TD * TJ >= 4 ? (REG16 ? 3 : 2) : 3)) void wgrad_gemm_kernel(const WgradArgs a) {
    constexpr int BD = 32 * TD * WD;
    constexpr int BJ = 32 * TJ * WJ;
    constexpr int DPT = BD / 8, JPT = BJ / 8;  // elements per thread per step
    static_assert(WD * WJ == 4, "4 waves");
    // ONE staging buffer (two barriers per 32-position step): half the LDS of a double buffer, so
    // 2-4 blocks share a CU and cover each other's barriers and global-load latency.
    __shared__ float Ds[32][BD + 1];
    __shared__ float Gs[32][BJ + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wave / WJ, wj = wave % WJ;
    const int tiles_d = a.DCp / BD;
    const int d_t = blockIdx.x % tiles_d, j_t = blockIdx.x / tiles_d;
    const int d0 = d_t * BD, j0 = j_t * BJ;
    const int ml = tid & 31, sub = tid >> 5;

    const int m_begin = blockIdx.y * a.chunk;
    const int m_end = min(a.M, m_begin + a.chunk);
    const int nit = (m_end > m_begin) ? (m_end - m_begin + 31) / 32 : 0;

    // all 32-bit byte offsets are relative to the sample of the block's first position; elements
    // outside the tensors (padding taps, ragged tails) get offset 0x80000000 -> hardware returns 0
    const uint32_t nb = fdiv((uint32_t)(nit > 0 ? m_begin : 0), a.div_sp);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dptr + (int64_t)nb * a.d_sn), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gptr + (int64_t)nb * a.g_sn), 0, 0x40000000u, 0x00020000);

    // this thread's J rows (gathered channel, tap) are fixed for the whole reduction
    int32_t goff4[REG16 ? 1 : JPT];
    uint32_t gsel[REG16 ? 1 : JPT];
    int32_t gso[REG16 ? JPT / 2 : 1];     // REG16: scalar (channel, depth-tap) offsets of row pairs
    uint32_t jmask = 0;                    // REG16: bit i = row j0 + sub + 8 i exists
    int32_t tapA = 0, tapB = 0;
    uint32_t selA = 0, selB = 0;
    if constexpr (REG16) {
        const int subu = sub & 7;
        tapA = a.hw_off4[subu]; tapB = a.hw_off4[subu + 8];
        selA = a.hw_sel[subu]; selB = a.hw_sel[subu + 8];
#pragma unroll
        for (int q = 0; q < JPT / 2; ++q) {
            const int cd = j0 / 16 + q;   // (sub + 8 i) / 16 == i / 2 for sub < 8
            gso[q] = (cd >> a.log2nd) * a.g_sc4 + (cd & ((1 << a.log2nd) - 1)) * a.g_sd4;
        }
#pragma unroll
        for (int i = 0; i < JPT; ++i)
            if (j0 + sub + 8 * i < a.J) jmask |= 1u << i;
    } else {
#pragma unroll
        for (int i = 0; i < JPT; ++i) {
            const int j = j0 + sub + 8 * i;
            if (j < a.J) {
                const KEntry e = a.jtab[j];
                goff4[i] = e.x_off * 4;
                gsel[i] = e.tapsel;
            } else {
                goff4[i] = 0;
                gsel[i] = 1u << 31;
            }
        }
    }
    const int dcb4 = (d0 + sub) * a.d_sc4;   // byte offset of this thread's first dense channel
    const int dstep4 = 8 * a.d_sc4;
    uint32_t dcmask = 0;                      // bit i: dense channel d0 + sub + 8 i exists
#pragma unroll
    for (int i = 0; i < DPT; ++i)
        if (d0 + sub + 8 * i < a.DC) dcmask |= 1u << i;

    f32x16 acc[TD][TJ];
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float dv[DPT], gv[JPT];

    auto load_tile = [&](int it) {
        const int m = m_begin + it * 32 + ml;
        const bool mok = m < m_end;
        const int mm = mok ? m : m_begin;
        const uint32_t n = fdiv((uint32_t)mm, a.div_sp);
        uint32_t r = (uint32_t)mm - n * a.div_sp.div;
        const uint32_t pd = fdiv(r, a.div_hw);
        r -= pd * a.div_hw.div;
        const uint32_t ph = fdiv(r, a.div_w);
        const uint32_t pw = r - ph * a.div_w.div;
        const int dbase4 = 4 * ((int)((int64_t)(n - nb) * a.d_sn) + (int)pd * a.d_sd + (int)ph * a.d_sh + (int)pw * a.d_sw) + dcb4;
        const uint32_t dm = mok ? dcmask : 0u;
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            const uint32_t vo = ((dm >> i) & 1u) ? (uint32_t)(dbase4 + i * dstep4) : 0x80000000u;
            dv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, vo, 0, 0));
        }
        const uint32_t vmask = mok ? (dim_mask(a.td, (int)pd, 0) | dim_mask(a.th, (int)ph, 8) | dim_mask(a.tw, (int)pw, 16)) : 0u;
        const int gbase4 = 4 * ((int)((int64_t)(n - nb) * a.g_sn) + ((int)pd * a.td.mul + a.td.base) * a.g_sd +
                                ((int)ph * a.th.mul + a.th.base) * a.g_sh + ((int)pw * a.tw.mul + a.tw.base) * a.g_sw);
        if constexpr (REG16) {
            const uint32_t voA = ((vmask & selA) == selA) ? (uint32_t)(gbase4 + tapA) : 0x80000000u;
            const uint32_t voB = ((vmask & selB) == selB) ? (uint32_t)(gbase4 + tapB) : 0x80000000u;
#pragma unroll
            for (int i = 0; i < JPT; ++i) {
                const uint32_t vo = ((jmask >> i) & 1u) ? ((i & 1) ? voB : voA) : 0x80000000u;
                gv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, vo, gso[i / 2], 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < JPT; ++i) {
                const uint32_t vo = ((vmask & gsel[i]) == gsel[i]) ? (uint32_t)(gbase4 + goff4[i]) : 0x80000000u;
                gv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, vo, 0, 0));
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < DPT; ++i) Ds[ml][sub + 8 * i] = dv[i];
#pragma unroll
        for (int i = 0; i < JPT; ++i) Gs[ml][sub + 8 * i] = gv[i];
    };

    if (nit > 0) {
        load_tile(0);
        store_tile();
    }
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    for (int it = 0; it < nit; ++it) {
        if (it + 1 < nit) load_tile(it + 1);   // global loads stay in flight under the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(2);
        float af[2][TD], bf[2][TJ];
#pragma unroll
        for (int i = 0; i < TD; ++i) af[0][i] = Ds[lhi][(wd * TD + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < TJ; ++j) bf[0][j] = Gs[lhi][(wj * TJ + j) * 32 + l31];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 16) {
                const int k = 2 * (ks + 1) + lhi;
#pragma unroll
                for (int i = 0; i < TD; ++i) af[nxt][i] = Ds[k][(wd * TD + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < TJ; ++j) bf[nxt][j] = Gs[k][(wj * TJ + j) * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < TD; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TD + TJ, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TD * TJ, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                        // every wave has read the tile
        if (it + 1 < nit) store_tile();
        __syncthreads();                        // next tile visible
    }

    float* __restrict__ out = a.slab + (int64_t)blockIdx.y * a.DCp * a.Jp;
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dc = d0 + (wd * TD + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int jj = j0 + (wj * TJ + j) * 32 + l31;
                out[(int64_t)dc * a.Jp + jj] = acc[i][j][r];
            }
}

// --------------------------------------------------------------------------- //
// wgrad GEMM, double-buffered LDS-DMA form (4x4 inner taps, 128 x 128 tile).
// Tiles are kept [channel row][64 positions] with a pitch of 65 words: every row is ONE
// lane-linear 256-B buffer_load...lds (positions on the lanes) and the MFMA fragment read
// "32 channels x one position" hits 32 different banks.  The per-lane voffset carries the
// position (for the gathered operand one of the 16 (kh,kw) taps, padding folded in as
// 0x80000000); channel and depth tap ride on the scalar soffset.  Two 66.5 KB buffers = one
// workgroup per CU, one wave per SIMD: while the 128 MFMAs of a tile run (8192 cycles), the
// next tile's 64 DMAs per wave are issued two per k-step in their shadow; one barrier per tile.
// --------------------------------------------------------------------------- //
// Per-lane voffsets of one 64-position tile (position m on this lane).  4x4 inner taps: the gathered
// operand's offset is plane + row(kh) + col(kw); an invalid row / column / position contributes
// 0x40000000 = num_records of the descriptor (block-relative offsets stay below it), so any sum with
// an invalid part is out of range and the DMA writes 0 — 8 range tests and 16 adds instead of 16
// mask tests, and straight-line code that the scheduler can spread under the MFMAs.
__device__ __forceinline__ void wgrad_tile_addr(const WgradArgs& a, int m, int m_begin, int m_end, uint32_t nb, uint32_t& dvo, uint32_t (&gvo)[16]) {
    const bool mok = m < m_end;
    const int mm = mok ? m : m_begin;
    const uint32_t n = fdiv((uint32_t)mm, a.div_sp);
    uint32_t r = (uint32_t)mm - n * a.div_sp.div;
    const uint32_t pd = fdiv(r, a.div_hw);
    r -= pd * a.div_hw.div;
    const uint32_t ph = fdiv(r, a.div_w);
    const uint32_t pw = r - ph * a.div_w.div;
    const int ns = (int)(n - nb);
    dvo = mok ? (uint32_t)(4 * (ns * (int)a.d_sn + (int)pd * a.d_sd + (int)ph * a.d_sh + (int)pw * a.d_sw)) : 0x80000000u;
    const uint32_t plane = mok ? (uint32_t)(4 * (ns * (int)a.g_sn + ((int)pd * a.td.mul + a.td.base) * a.g_sd)) : 0x40000000u;
    const int h0 = (int)ph * a.th.mul + a.th.base, w0 = (int)pw * a.tw.mul + a.tw.base;
    uint32_t row[4], col[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int hp = h0 + u, wp = w0 + u;
        row[u] = plane + ((unsigned)hp < (unsigned)a.th.size ? (uint32_t)(hp * a.g_sh * 4) : 0x40000000u);
        col[u] = (unsigned)wp < (unsigned)a.tw.size ? (uint32_t)(wp * a.g_sw * 4) : 0x40000000u;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) gvo[t] = row[t >> 2] + col[t & 3];
}

// TD = dense-operand row tiles per wave: 2 -> 128 x 128 tile, 1 -> 64 x 128 (64-channel layers: cgen.down0, gdis.5)
// D16 (round 3): the dense operand's rows — contiguous, 16-byte aligned runs of positions — are staged by 16-byte LDS-DMAs, FOUR rows per
// instruction (lanes 16 i .. 16 i + 15 fetch row i) instead of one row per dword DMA: DR / 4 dense DMAs per wave and tile instead of DR (40 instead
// of 64 DMAs per 128 x 128 tile, 36 instead of 48 for the 64-row tile).  An LDS-DMA lands lane-linear, so the four rows of an instruction sit at a
// pitch of 64 words — the same banks; lane (row i, granule q) therefore fetches granule q ^ i of its row (an XOR swizzle on the global side),
// and row groups are 272 words apart: a fragment read "32 rows x one position" then has a 2-way bank conflict at worst (16 B granules leave 16 of the
// 64 banks per word phase), which costs ~2 cycles against the 64 of the MFMA it feeds.
template <int TD, int BF = 0, bool D16 = false>   // BF: 0 fp32 MFMA, 1 bf16 products, 2 fp32 on the bf16 pipe (both operands split three ways in registers)
__global__ __launch_bounds__(256, 1) void wgrad_dma_kernel(const WgradArgs a) {
    static_assert(!(BF && D16), "the 16-byte dense staging is built for the fp32 form");
    constexpr int BD = 64 * TD, BJ = 128, P = 65, GP = 272, DREG = D16 ? (BD / 4) * GP : BD * P, TILE = DREG + BJ * P, DR = 16 * TD;   // DR: dense rows DMA'd per wave
    __shared__ __attribute__((aligned(16))) float smem[2 * TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wave >> 1, wj = wave & 1;
    const int tiles_d = a.DCp / BD;
    const int d_t = blockIdx.x % tiles_d, j_t = blockIdx.x / tiles_d;
    const int d0 = d_t * BD, j0 = j_t * BJ;
    const int m_begin = blockIdx.y * a.chunk;
    const int m_end = min(a.M, m_begin + a.chunk);
    const int nit = (m_end > m_begin) ? (m_end - m_begin + 63) / 64 : 0;

    const uint32_t nb = fdiv((uint32_t)(nit > 0 ? m_begin : 0), a.div_sp);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dptr + (int64_t)nb * a.d_sn), 0, 0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gptr + (int64_t)nb * a.g_sn), 0, 0x40000000u, 0x00020000);

    // scalar row offsets of this wave's 32 + 32 rows (loop invariant)
    const int cd0 = (j0 >> 4) + wave * 2;   // (channel, depth tap) index of rows 0..15; rows 16..31 -> cd0 + 1
    const int gso0 = (cd0 >> a.log2nd) * a.g_sc4 + (cd0 & ((1 << a.log2nd) - 1)) * a.g_sd4;
    const int gso1 = ((cd0 + 1) >> a.log2nd) * a.g_sc4 + ((cd0 + 1) & ((1 << a.log2nd) - 1)) * a.g_sd4;
    const int dso0 = (d0 + wave * DR) * a.d_sc4;

    // two-level accumulation (registers are free at one wave per SIMD): the MFMA adds into `acc` as a
    // k-ordered fp32 chain; every 16 tiles (1024 positions) the chain is folded into `sum`, so no
    // partial sum is longer than 1024 terms before it meets a value of its own magnitude
    f32x16 acc[TD][2], sum[TD][2];
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; sum[i][j][r] = 0.f; }

    uint32_t dvo = 0x80000000u, gvo[16], dvon = 0x80000000u, gvon[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { gvo[t] = 0x80000000u; gvon[t] = 0x80000000u; }
#define DCV_WG_ADDR(IT, DVO, GVO) wgrad_tile_addr(a, m_begin + (IT) * 64 + lane, m_begin, m_end, nb, DVO, GVO);
#if defined(__HIP_DEVICE_COMPILE__)
    // row I (0..31) of this wave's dense / gathered share of the tile in buffer BUF
#define DCV_WG_DROW(BUF, I) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void*)(smem + (BUF) * TILE + (wave * DR + (I)) * P), 4, dvo, dso0 + (I) * a.d_sc4, 0, 0);
    // D16: rows 4 Q .. 4 Q + 3 of this wave's dense share in one instruction (dvo holds the lane's swizzled granule + row offset)
#define DCV_WG_DGRP(BUF, Q) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void*)(smem + (BUF) * TILE + (wave * (DR / 4) + (Q)) * GP), 16, dvo, dso0 + 4 * (Q) * a.d_sc4, 0, 0);
#define DCV_WG_GROW(BUF, I) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (lds_void*)(smem + (BUF) * TILE + DREG + (wave * 32 + (I)) * P), 4, gvo[(I) & 15], ((I) < 16 ? gso0 : gso1), 0, 0);
#else
#define DCV_WG_DROW(BUF, I) { (void)dso0; }
#define DCV_WG_DGRP(BUF, Q) { (void)dso0; }
#define DCV_WG_GROW(BUF, I) { (void)gso0; (void)gso1; }
#endif
    // D16: lane (i = lane / 16, q = lane % 16) stages granule q ^ i of row i of every 4-row group: its voffset is the dword form's voffset of
    // position 4 (q ^ i) of the tile (a cross-lane read; M, the chunk and the row length are multiples of 4 and rows are 16-byte aligned —
    // checked on the host — so a granule is valid or padding as a whole) plus i channel strides
    const int d16_src = 4 * (4 * ((lane & 15) ^ (lane >> 4)));          // ds_bpermute byte address of the source lane
    const uint32_t d16_row = (uint32_t)(lane >> 4) * (uint32_t)a.d_sc4;
#define DCV_WG_D16(DVO) if constexpr (D16) { DVO = (uint32_t)__builtin_amdgcn_ds_bpermute(d16_src, (int)(DVO)); DVO = (DVO & 0x80000000u) ? 0x80000000u : DVO + d16_row; }

    const int l31 = lane & 31, lhi = lane >> 5;
    if (nit > 0) {
        DCV_WG_ADDR(0, dvo, gvo)
        DCV_WG_D16(dvo)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if constexpr (D16) { if (i < DR / 4) { DCV_WG_DGRP(0, i < DR / 4 ? i : 0) } }
            else { if (i < DR) { DCV_WG_DROW(0, i < DR ? i : 0) } }
            DCV_WG_GROW(0, i)
        }
        DCV_WG_ADDR(min(1, nit - 1), dvo, gvo)   // voffsets of the tile whose DMAs the first loop step issues
        DCV_WG_D16(dvo)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef DCV_STAMP
    unsigned long long s_mfma = 0, s_addr = 0, s_wait = 0, s_t0 = clock64(), s_r0 = __builtin_amdgcn_s_memrealtime();   // shader clock; constant 100 MHz clock
#endif
    for (int it = 0; it < nit; ++it) {
        const int buf = it & 1;
        STAMP(w0);
        // dvo / gvo hold tile min(it + 1, nit - 1) (last tile: a harmless repeat into the idle buffer)
        const float* da = smem + buf * TILE + ((wd * TD) * 32 + l31) * P + lhi;
        const float* gb = smem + buf * TILE + DREG + ((wj * 2) * 32 + l31) * P + lhi;
        // D16: row r = (wd TD + i) 32 + l31 lives in group r / 4 at row slot r % 4 = l31 % 4; position p = 2 ks + lhi of it at word
        // (((p >> 2) ^ (l31 & 3)) << 2) | (p & 3): one base per value of (ks >> 1) & 3, everything else is an immediate
        const float* dq[4];
        if constexpr (D16) {
#pragma unroll
            for (int lo = 0; lo < 4; ++lo)
                dq[lo] = smem + buf * TILE + ((wd * TD) * 8 + (l31 >> 2)) * GP + (l31 & 3) * 64 + ((lo ^ (l31 & 3)) << 2) + lhi;
        } else {
            dq[0] = dq[1] = dq[2] = dq[3] = da;
        }
#define DCV_WG_AF(I, KS) (D16 ? dq[((KS) >> 1) & 3][(I) * 8 * GP + (((KS) >> 1) & ~3) * 4 + 2 * ((KS) & 1)] : da[(I) * 32 * P + 2 * (KS)])
        __builtin_amdgcn_s_setprio(2);
        // voffsets of the tile after next: plain VALU work with no consumer inside this step, free to be
        // scheduled into the shadow of the MFMAs below (one wave per SIMD: nothing else would hide it)
        DCV_WG_ADDR(min(it + 2, nit - 1), dvon, gvon)
        DCV_WG_D16(dvon)
        if constexpr (BF) {
            // bf16 products: the tile's 64 positions are 4 MFMA k-blocks of 16; a lane of half-wave lhi holds positions
            // 16 s + 8 lhi .. + 7 of its channel row.  The next tile's DR + 32 row DMAs go out first.
#pragma unroll
            for (int q = 0; q < 32; ++q) { if (q < DR) { DCV_WG_DROW(buf ^ 1, q < DR ? q : 0) } DCV_WG_GROW(buf ^ 1, q) }
            // one base register per 32-row block, opaque to the optimiser: every fragment word is then base + a small immediate
            // (folded into ONE base the row-block offsets exceed ds_read2's 8-bit offset field and each read got its own v_add)
            uint32_t dao[2] = {(uint32_t)(da - smem) + 7 * lhi, (uint32_t)(da - smem) + 7 * lhi + (TD > 1 ? 32 * P : 0)};
            uint32_t gbo[2] = {(uint32_t)(gb - smem) + 7 * lhi, (uint32_t)(gb - smem) + 7 * lhi + 32 * P};
            asm volatile("" : "+v"(dao[0]), "+v"(dao[1]), "+v"(gbo[0]), "+v"(gbo[1]));
            const float* dab[2] = {smem + dao[0], smem + dao[1]};
            const float* gbb[2] = {smem + gbo[0], smem + gbo[1]};
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) {
                if constexpr (BF == 2) {
                    bf16x8 a3[TD][3], b3[2][3];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float t[8];
                        if (i < TD) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) t[q] = dab[i][16 * sb + q];
                            split3_bf16x8(t, a3[i < TD ? i : 0]);
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) t[q] = gbb[i][16 * sb + q];
                        split3_bf16x8(t, b3[i]);
                    }
#pragma unroll
                    for (int i = 0; i < TD; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) DCV_MFMA_X6(acc[i][j], a3[i], b3[j])
                } else {
                bf16x8 a8[TD], b8[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float t[8];
                    if (i < TD) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) t[q] = dab[i][16 * sb + q];
                        a8[i < TD ? i : 0] = pack_bf16x8(t);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = gbb[i][16 * sb + q];
                    b8[i] = pack_bf16x8(t);
                }
#pragma unroll
                for (int i = 0; i < TD; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
                }
            }
        } else {
        float af[2][TD], bf[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { if (i < TD) af[0][i < TD ? i : 0] = DCV_WG_AF(i, 0); bf[0][i] = gb[i * 32 * P]; }
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 32) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { if (i < TD) af[nxt][i < TD ? i : 0] = DCV_WG_AF(i, ks + 1); bf[nxt][i] = gb[i * 32 * P + 2 * (ks + 1)]; }
            }
#pragma unroll
            for (int i = 0; i < TD; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            // next tile's DR + 32 row DMAs, three per k-step (dense rows first interleaved with gathered ones): the last
            // one is issued about ten k-steps (2500 cycles) before the wait at the end of the tile
            if constexpr (D16) {
                // DR / 4 + 32 DMAs: two per k-step (the dense groups first, each followed by a gathered row), the last one ~12 k-steps before the wait
#pragma unroll
                for (int q = 2 * ks; q < 2 * ks + 2; ++q)
                    if (q < DR / 4 + 32) {
                        if (q < DR / 2) {
                            if (q & 1) { DCV_WG_GROW(buf ^ 1, q >> 1) } else { DCV_WG_DGRP(buf ^ 1, (q >> 1) < DR / 4 ? (q >> 1) : 0) }
                        } else {
                            DCV_WG_GROW(buf ^ 1, (q - DR / 4) < 32 ? (q - DR / 4) : 0)
                        }
                    }
            } else {
#pragma unroll
            for (int q = 3 * ks; q < 3 * ks + 3; ++q)
                if (q < DR + 32) {
                    if (q < 2 * DR) {
                        if (q & 1) { DCV_WG_GROW(buf ^ 1, q >> 1) } else { DCV_WG_DROW(buf ^ 1, (q >> 1) < DR ? (q >> 1) : 0) }
                    } else {
                        DCV_WG_GROW(buf ^ 1, (q - DR) < 32 ? (q - DR) : 0)
                    }
                }
            }
            // issue order of a k-step: every MFMA is followed by a share of the step's other instructions — the wave issues in order
            // and an MFMA occupies the matrix pipe for 64 cycles, so whatever follows ONE MFMA issues in its shadow for free, while
            // the step's 2 + TD fragment reads and 3 row DMAs (~90 issue cycles) all behind the LAST MFMA overran its shadow by ~25
            // cycles (8810 -> 8582 cycles per 128-MFMA tile, 8192 being the matrix pipe's own time)
            if constexpr (D16) {
                constexpr int NI = DR / 4 + 32;
                const int left = NI - 2 * ks < 0 ? 0 : (NI - 2 * ks > 2 ? 2 : NI - 2 * ks);
                if constexpr (TD == 2) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (left >= 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (left == 2) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    if (left >= 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (left == 2) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            } else {
                constexpr int NI = DR + 32;
                const int left = NI - 3 * ks < 0 ? 0 : (NI - 3 * ks > 3 ? 3 : NI - 3 * ks);
                if constexpr (TD == 2) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (left >= 2) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
                    else if (left == 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (left == 3) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    if (left >= 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (left == 3) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
                    else if (left == 2) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
        }
        }
        __builtin_amdgcn_s_setprio(0);
        STAMP(w1);
        if ((it & 15) == 15) {
#pragma unroll
            for (int i = 0; i < TD; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { sum[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
        }
        dvo = dvon;
#pragma unroll
        for (int t = 0; t < 16; ++t) gvo[t] = gvon[t];
        STAMP(w2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMAs of the next tile have landed
        __syncthreads();                                   // ... everyone's; and all reads of `buf` are done
#ifdef DCV_STAMP
        { unsigned long long w3 = clock64(); s_mfma += w1 - w0; s_addr += w2 - w1; s_wait += w3 - w2; }
#endif
    }
#ifdef DCV_STAMP
    if (lane == 0 && blockIdx.x + blockIdx.y * gridDim.x < 4096) {
        unsigned long long* g = g_stamp[blockIdx.x + blockIdx.y * gridDim.x][wave];
        g[0] = s_mfma; g[1] = s_addr; g[2] = s_wait; g[3] = __builtin_amdgcn_s_memrealtime() - s_r0; g[4] = clock64() - s_t0; g[5] = (unsigned long long)nit;
    }
#endif

    float* __restrict__ out = a.slab + (int64_t)blockIdx.y * a.DCp * a.Jp;
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dc = d0 + (wd * TD + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int jj = j0 + (wj * 2 + j) * 32 + l31;
                out[(int64_t)dc * a.Jp + jj] = sum[i][j][r] + acc[i][j][r];
            }
}

// --------------------------------------------------------------------------- //
// Thin weight gradient of the colour generator's stem (1 or 2 gathered channels -> 64, 3x3, 64-wide rows).  As a GEMM the
// reduction runs over positions and one side is only GC x 9 = 9 / 18 wide: the MFMA tile is mostly padding, the im2col staging
// of the thin operand dominates and the 1.2 GB dense gradient streams at a quarter of the HBM rate.  Here a lane owns one column
// of the dense tensor: the 3x3 neighbourhood of the thin operand lives in registers (rows loaded once, horizontal neighbours
// from DPP wave shifts), each dense value is loaded once (coalesced rows) and multiplied into DCW x GC x 9 per-lane
// accumulators; a wave walks `pps` images, then reduces its accumulators over the 64 lanes (DPP) and writes one slab row, which
// wgrad_reduce_kernel sums in a fixed order.  (Measured and dropped: the same scheme for 3 gathered channels - the RGB head,
// 0.89 ms against 0.60 ms on the MFMA path - and for the 4x4(x4) stride-2 stems, 0.18-0.48 ms against 0.13-0.24 ms: with 27+
// taps per dense value the MFMA tile is no longer mostly padding and the VALU form is slower.)
// --------------------------------------------------------------------------- //
struct ThinWgradArgs {
    const float* d;
    const float* g;
    float* slab;
    int32_t P, DC, OH, J, pps, pad;   // P images
    int64_t d_sn, d_sc, d_sh;
    int64_t g_sn, g_sc, g_sh;
};

__device__ __forceinline__ float wave_total(float v) {   // sum over the 64 lanes, wave-uniform
    v = half_wave_sum(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float dpp_prev_lane(float v) {   // lane - 1 (0 into lane 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_next_lane(float v) {   // lane + 1 (0 into lane 63)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// 3x3, stride 1, padding 1, 64-wide rows, 2-D: dw[dc][gc][kh][kw] = sum d[n, dc, r, c] * g[n, gc, r - 1 + kh, c - 1 + kw].
// The rows of the wave's images form one stream, walked in chunks of CH rows: the loads of the next chunk (CH dense rows of DCW
// channels, CH gathered rows of NG channels) are issued before the multiply-adds of the current one, and nothing in the loop
// is conditional (rows past an image's edge are loaded from a clamped address and zeroed by a select), so the waits are
// counter waits on loads issued a chunk earlier.
template <int NG, int DCW>
__global__ __launch_bounds__(256) void thin_wgrad3_kernel(const ThinWgradArgs a) {
    constexpr int CNT = DCW * NG * 9, CH = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dc0 = ((int)blockIdx.x * 4 + wave) * DCW;
    const int p0 = (int)blockIdx.y * a.pps, p1 = min(p0 + a.pps, a.P);
    const int H = a.OH;
    float acc[DCW][NG][3][3];
#pragma unroll
    for (int j = 0; j < DCW; ++j)
#pragma unroll
        for (int gc = 0; gc < NG; ++gc)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[j][gc][t / 3][t % 3] = 0.f;
    const float* __restrict__ gb = a.g + lane;
    const float* __restrict__ db = a.d + (int64_t)dc0 * a.d_sc + lane;
    float gn[CH][NG], dn[CH][DCW];
    // chunk (n, r0): dense rows r0 .. r0 + CH - 1 and gathered rows r0 + 1 .. r0 + CH of image n; the row after an image's last
    // one is row 0 of the next image (it becomes the window's centre row there; as a "row below" it is masked)
    auto load_chunk = [&](int n, int r0) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            int gr = r0 + i + 1, gi = n;
            if (gr >= H) { gr = 0; gi = min(n + 1, p1 - 1); }   // wave-uniform
            const float* gp = gb + (int64_t)gi * a.g_sn + (int64_t)gr * a.g_sh;
#pragma unroll
            for (int gc = 0; gc < NG; ++gc) gn[i][gc] = gp[(int64_t)gc * a.g_sc];
            const float* dp = db + (int64_t)n * a.d_sn + (int64_t)(r0 + i) * a.d_sh;
#pragma unroll
            for (int j = 0; j < DCW; ++j) dn[i][j] = dp[(int64_t)j * a.d_sc];
        }
    };
    float win[3][NG][3];   // gathered rows r - 1, r, r + 1; columns c - 1, c, c + 1
#pragma unroll
    for (int gc = 0; gc < NG; ++gc) {
        const float v = gb[(int64_t)p0 * a.g_sn + (int64_t)gc * a.g_sc];
        win[0][gc][0] = win[0][gc][1] = win[0][gc][2] = 0.f;
        win[1][gc][0] = dpp_prev_lane(v); win[1][gc][1] = v; win[1][gc][2] = dpp_next_lane(v);
    }
    load_chunk(p0, 0);
    int n = p0, r0 = 0;
    const int chunks = (p1 - p0) * (H / CH);
    for (int ch = 0; ch < chunks; ++ch) {
        float gq[CH][NG], dq[CH][DCW];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
#pragma unroll
            for (int gc = 0; gc < NG; ++gc) gq[i][gc] = gn[i][gc];
#pragma unroll
            for (int j = 0; j < DCW; ++j) dq[i][j] = dn[i][j];
        }
        int n2 = n, r2 = r0 + CH;
        if (r2 >= H) { r2 = 0; n2 = min(n + 1, p1 - 1); }
        load_chunk(n2, r2);   // the prefetch after the last chunk re-reads valid rows and is dropped
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const bool top = r0 + i == 0, bot = r0 + i == H - 1;
#pragma unroll
            for (int gc = 0; gc < NG; ++gc) {
                const float v = gq[i][gc];
                win[2][gc][0] = dpp_prev_lane(v); win[2][gc][1] = v; win[2][gc][2] = dpp_next_lane(v);
            }
#pragma unroll
            for (int gc = 0; gc < NG; ++gc)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float w0 = top ? 0.f : win[0][gc][kw], w1 = win[1][gc][kw], w2 = bot ? 0.f : win[2][gc][kw];
#pragma unroll
                    for (int j = 0; j < DCW; ++j) {
                        acc[j][gc][0][kw] += dq[i][j] * w0;
                        acc[j][gc][1][kw] += dq[i][j] * w1;
                        acc[j][gc][2][kw] += dq[i][j] * w2;
                    }
                }
#pragma unroll
            for (int gc = 0; gc < NG; ++gc)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) { win[0][gc][kw] = win[1][gc][kw]; win[1][gc][kw] = win[2][gc][kw]; }
        }
        n = n2; r0 = r2;
    }
    // accumulator idx = (j * NG + gc) * 9 + t is element dc0 * J + idx of the slab row (J = NG * 9): lane idx % 64 keeps it
    float out[(CNT + 63) / 64];
#pragma unroll
    for (int o = 0; o < (CNT + 63) / 64; ++o) out[o] = 0.f;
#pragma unroll
    for (int idx = 0; idx < CNT; ++idx) {
        const float tot = wave_total(acc[idx / (NG * 9)][idx / 9 % NG][idx % 9 / 3][idx % 3]);
        if (lane == idx % 64) out[idx / 64] = tot;
    }
    float* __restrict__ row = a.slab + (int64_t)blockIdx.y * a.DC * a.J + (int64_t)dc0 * a.J;
#pragma unroll
    for (int o = 0; o < (CNT + 63) / 64; ++o)
        if (o * 64 + lane < CNT) row[o * 64 + lane] = out[o];
}

// dw[dc][j] = sum_s slab[s][dc][j].  64 outputs per block; the 4 waves each sum every 4th split
// (independent loads, 4-way unrolled) and wave 0 combines the four partial sums in a fixed order:
// bitwise reproducible, and 16x more loads in flight than one thread walking all S splits.
// J % 4 == 0 (every 4x4-tap layer): four consecutive j per lane, 16-byte loads — the same order of additions per element
// --------------------------------------------------------------------------- //
// thinj_wgrad_kernel (round 6): the weight gradient of a 3x3 / stride-1 / pad-1 layer with at most 3 GATHERED channels and 128 k dense ones on 64-wide
// rows — the colour generator's RGB head, R[dc][gc][kh][kw] = sum D[n, dc, r, c] G[n, gc, r - 1 + kh, c - 1 + kw] with D the head's 2.35 GB input.  The
// generic tile (128 x 32, one LDS buffer, two barriers per 32 positions) ran it at 0.61 ms = 3.9 TB/s; the op is J = 9 GC <= 32 wide, so one 32x32 MFMA
// column covers it and the kernel only has to stream D once.  A wave owns 32 dense channels and ONE accumulator tile for its whole life; its A operand is
// read straight from HBM, 16 bytes per lane: lane (channel i, half h) holds columns 8q + 4h .. + 3 of a row, and MFMA (q, e) reduces over columns
// 8q + e (half 0) and 8q + 4 + e (half 1) — any pairing of positions is a valid K order as long as B follows it.  B (the gathered taps of the same two
// columns, lane = (gc, kh, kw)) comes from the chunk's G rows staged in LDS ([row][gc][x + 1], pitch 67: the 27 tap addresses of a half wave fall
// on 27 different banks).  The four waves of a workgroup share the position chunk and the staged rows; a workgroup walks `cpw` chunks and writes one slab.
// --------------------------------------------------------------------------- //
struct ThinJArgs {
    const float* d;
    const float* g;
    float* slab;
    int32_t OH, DC, GC, J, rpc, nchunk, cpw, cpi;   // rows per chunk, chunks, chunks per workgroup, chunks per image
    int64_t d_sn, g_sn;
    int32_t d_sc, d_sh, g_sc, g_sh;
};

template <int GC, bool BN = false>
__global__ __launch_bounds__(256, 3) void thinj_wgrad_kernel(const ThinJArgs a, const BnView bv = BnView()) {
    constexpr int PITCH = 67, RMAX = 8 + 2;
    __shared__ float gs[2][RMAX * GC][PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int rpc = a.rpc, nrow = (rpc + 2) * GC;
    for (int e = tid; e < 2 * RMAX * GC * PITCH; e += 256) (&gs[0][0][0])[e] = 0.f;   // halo columns (and everything a lane past J may read) are zero for good
    const int c0 = blockIdx.x * a.cpw, c1 = min(a.nchunk, c0 + a.cpw);
    const int ch = blockIdx.y * 128 + wave * 32 + l31;                                  // this lane's dense channel
    const uint32_t dvo = ch < a.DC ? (uint32_t)(4 * (ch * a.d_sc + 4 * lhi)) : 0x80000000u;
    // this lane's tap: j = (gc, kh, kw); lanes past J read tap 0 (their columns are not stored)
    const int j = l31 < a.J ? l31 : 0;
    const int jgc = j / 9, jt = j - jgc * 9, jkh = jt / 3, jkw = jt - jkh * 3;
    const int jbase = (jkh * GC + jgc) * PITCH + jkw + 4 * lhi;                           // word offset of column 0 of chunk row 0 for this lane's tap
    // staging: element e of a chunk's G rows = (row rr of nrow, column x): thread t takes e = t, t + 256, ..
    constexpr int SPT = (RMAX * GC * 64 + 255) / 256;
    float sv[SPT];
    auto stage_load = [&](int c) {
        const int n = c / a.cpi, y0 = (c - n * a.cpi) * rpc;
        const float* __restrict__ gb = a.g + (int64_t)n * a.g_sn;
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int e = tid + 256 * i, rr = e >> 6, x = e & 63;
            const int yl = rr / GC, gc = rr - yl * GC, y = y0 - 1 + yl;
            sv[i] = (rr < nrow && (unsigned)y < (unsigned)a.OH) ? gb[(int64_t)gc * a.g_sc + (int64_t)y * a.g_sh + x] : 0.f;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int e = tid + 256 * i, rr = e >> 6, x = e & 63;
            if (rr < nrow) gs[buf][rr][x + 1] = sv[i];
        }
    };
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 av[2][8];
    // BN: this wave's 32 dense channels are a BatchNorm group's output that was never written: read its INPUT, normalise + activate per element (wave-uniform choice:
    // cbn is a multiple of 32); the channel's constants live in the lane
    [[maybe_unused]] const bool bnw = BN && __builtin_amdgcn_readfirstlane(ch - l31) < bv.cbn;
    [[maybe_unused]] float bsc = 1.f, bsh = 0.f;
    [[maybe_unused]] uint32_t bvo = 0x80000000u;
    if constexpr (BN) {
        if (bnw) {
            const float is = bv.invstd[ch];
            bsc = bv.gamma[ch] * is;
            bsh = bv.beta[ch] - bv.mean[ch] * bsc;
            bvo = (uint32_t)(4 * (ch * bv.bx_sc + 4 * lhi));
        }
    }
    auto load_a = [&](int c, int rl, f32x4 (&dst)[8]) {
        const int n = c / a.cpi, y = (c - n * a.cpi) * rpc + rl;
        if (BN && bnw) {
            const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bv.bx + (int64_t)n * bv.bx_sn + (int64_t)y * bv.bx_sh), 0, 0x80000000u, 0x00020000);
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, bvo, 32 * q, 0));
            return;
        }
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d + (int64_t)n * a.d_sn + (int64_t)y * a.d_sh), 0, 0x80000000u, 0x00020000);
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, dvo, 32 * q, 0));
    };
    [[maybe_unused]] auto bn_fix = [&](float v) {
        const float z = v * bsc + bsh;
        return bv.act == DCV_ACT_LEAKY ? (z > 0.f ? z : z * bv.slope) : z;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __syncthreads();
    if (c0 < c1) {
        stage_load(c0);
        stage_store(0);
        load_a(c0, 0, av[0]);
    }
    __syncthreads();
    int buf = 0;
    for (int c = c0; c < c1; ++c, buf ^= 1) {
        if (c + 1 < c1) stage_load(c + 1);
        for (int rl = 0; rl < rpc; rl += 2) {
            // two rows per trip so the A double buffer is indexed statically
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = rl + u;
                if (r >= rpc) break;
                // next row's A (the next chunk's first row at the end of this one)
                if (r + 1 < rpc) load_a(c, r + 1, av[u ^ 1]);
                else if (c + 1 < c1) load_a(c + 1, 0, av[u ^ 1]);
                const float* __restrict__ brow = &gs[buf][0][0] + jbase + r * (GC * PITCH);
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float av_ = av[u][q][e];
                        if (BN && bnw) av_ = bn_fix(av_);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av_, brow[8 * q + e], acc, 0, 0, 0);
                    }
            }
        }
        if (c + 1 < c1) stage_store(buf ^ 1);
        __syncthreads();
    }
    float* __restrict__ out = a.slab + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 128 * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dcl = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        out[dcl * 32 + l31] = l31 < a.J ? acc[r] : 0.f;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int DC, int J, int DCp, int Jp, int acc) {
    __shared__ float4 part[3][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int J4 = J >> 2;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const bool ok = i < (int64_t)DC * J4;
    const int dc = ok ? (int)(i / J4) : 0, j = ok ? (int)(i % J4) * 4 : 0;
    const int64_t stride = ((int64_t)DCp * Jp) >> 2;
    const float4* p = reinterpret_cast<const float4*>(slab + (int64_t)dc * Jp + j);
    float4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    int k = grp;
    for (; k + 12 < S; k += 16) {
        const float4 t0 = p[(int64_t)k * stride], t1 = p[(int64_t)(k + 4) * stride], t2 = p[(int64_t)(k + 8) * stride], t3 = p[(int64_t)(k + 12) * stride];
        add(s0, t0); add(s1, t1); add(s2, t2); add(s3, t3);
    }
    for (; k < S; k += 4) add(s0, p[(int64_t)k * stride]);
    const float4 v = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
    if (grp > 0) part[grp - 1][lane] = v;
    __syncthreads();
    if (grp == 0 && ok) {
        const float4 a = part[0][lane], b = part[1][lane], c = part[2][lane];
        float4 r = float4{((v.x + a.x) + b.x) + c.x, ((v.y + a.y) + b.y) + c.y, ((v.z + a.z) + b.z) + c.z, ((v.w + a.w) + b.w) + c.w};
        float4* o = reinterpret_cast<float4*>(dw + (int64_t)dc * J + j);
        if (acc) { const float4 old = *o; r = float4{old.x + r.x, old.y + r.y, old.z + r.z, old.w + r.w}; }      // = the sum autograd would form of the two gradients (one rounding, commutative)
        *o = r;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int DC, int J, int DCp, int Jp, int acc) {
    __shared__ float part[3][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const bool ok = i < (int64_t)DC * J;
    const int dc = ok ? (int)(i / J) : 0, j = ok ? (int)(i % J) : 0;
    const int64_t stride = (int64_t)DCp * Jp;
    const float* p = slab + (int64_t)dc * Jp + j;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = grp;
    for (; k + 12 < S; k += 16) {
        s0 += p[(int64_t)k * stride];
        s1 += p[(int64_t)(k + 4) * stride];
        s2 += p[(int64_t)(k + 8) * stride];
        s3 += p[(int64_t)(k + 12) * stride];
    }
    for (; k < S; k += 4) s0 += p[(int64_t)k * stride];
    const float v = (s0 + s1) + (s2 + s3);
    if (grp > 0) part[grp - 1][lane] = v;
    __syncthreads();
    if (grp == 0 && ok) {
        const float r = ((v + part[0][lane]) + part[1][lane]) + part[2][lane];
        dw[i] = acc ? dw[i] + r : r;
    }
}

// Many slabs, few elements (stem3d_wgrad_kernel: ~1000-2000 slabs of 2-6 k elements): wgrad_reduce4_kernel gives an element one column of FOUR threads, i.e. S / 4
// dependent-latency loads per thread on 8-24 workgroups — 20-25 us, a third of the stem kernel's own time.  Here a float4 of elements gets 64 threads, each with four
// independent partial sums over the slabs g, g + 64, ...; the 64 partials meet in LDS and are added in group order by one thread.  Fixed order: bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int E4, int acc) {      // E4 float4 elements per slab, dw [4 E4]
    constexpr int EPB = 4, G = 64;
    __shared__ float4 part[G][EPB];
    const int e = threadIdx.x & (EPB - 1), grp = threadIdx.x >> 2;
    const int64_t i = (int64_t)blockIdx.x * EPB + e;
    const bool ok = i < E4;
    const float4* __restrict__ p = reinterpret_cast<const float4*>(slab) + (ok ? i : 0);
    float4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    if (ok) {
        int k = grp;
        for (; k + 3 * G < S; k += 4 * G) {
            const float4 t0 = p[(int64_t)k * E4], t1 = p[(int64_t)(k + G) * E4], t2 = p[(int64_t)(k + 2 * G) * E4], t3 = p[(int64_t)(k + 3 * G) * E4];
            add(s0, t0); add(s1, t1); add(s2, t2); add(s3, t3);
        }
        for (; k < S; k += G) add(s0, p[(int64_t)k * E4]);
    }
    part[grp][e] = float4{(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
    __syncthreads();
    if (grp == 0 && ok) {
        float4 r = part[0][e];
        for (int g = 1; g < G; ++g) add(r, part[g][e]);
        float4* o = reinterpret_cast<float4*>(dw) + i;
        if (acc) { const float4 old = *o; r = float4{old.x + r.x, old.y + r.y, old.z + r.z, old.w + r.w}; }      // = the sum autograd would form of the two gradients (one rounding)
        *o = r;
    }
}

// --------------------------------------------------------------------------- //
// stem3d_wgrad_kernel<CIN>: weight gradient of the 3-D discriminators' stems — Conv3d(CIN -> 32, 4x4x4, stride (1,2,2), padding (0,1,1)) on 64-wide frames
// (discriminator.py: the depth / flow and colour branches of VideoDiscriminator, GradientDiscriminator's first layer; CIN = 1, 2, 3), R[oc][j] = sum_m dy[oc][m]
// xg[j][m] with J = 64 CIN rows j = (c, kd, kh, kw).  The generic tile kernel runs it as a 32 x 128 / 32 x 256 tile over ~970 position slabs: half of the tile is
// padding for one input channel, every input element is gathered 16 times through the texture path, 0.16-0.32 of the MFMA rate — and the iteration pays those nine
// launches' 1.5 ms nearly one for one (profiles/r06_ab_small_experiments.txt).  Here ONE WAVE owns a run of output rows (n, od, oh) and all J columns:
//   * per output row it stages, by wave-private 16-byte LDS-DMA one row ahead, the dy row of the 32 channels (4 instructions: 8 channels x 8 granules each) and the 4
//     input rows 2 oh - 1 .. 2 oh + 2 of each of the 4 CIN (channel, depth tap) planes (ONE instruction per plane: 4 rows x 16 granules; rows outside the frame
//     arrive as zeros).  An LDS-DMA lands lane-linear, so both images are permuted on the SOURCE side: dy granule g of channel oc sits at g ^ ((oc >> 1) & 7) — the
//     A fragment "32 channels x four positions" is then a conflict-free ds_read_b128 that serves two K steps; x granule c of row kh of plane pl sits at
//     c ^ (kh + 4 (pl & 1)) — the B fragment, a lane per tap (kd & 1, kh, kw) at one position, reads 32 different banks (the first version's dword DMAs, 16 + 16 CIN per
//     row at ~90 cycles of issue each, cost as much as the row's MFMAs);
//   * a K step is two neighbouring positions; the taps that fall on the column halo (kw = 0 at the first position, kw = 3 at the last) are zeroed in the two steps
//     where they occur; the step's B addresses are one XOR away from lane constants, the column tile is an immediate; 2 CIN MFMAs per step share the dy fragment;
//   * no workgroup barrier anywhere (a workgroup IS a wave: its own vmcnt orders the DMAs before its reads), 16-32 KB of LDS: 4-8 waves per CU;
//   * partial sums leave as one slab per wave, [32][J] = dw's own layout; wgrad_reduce4_kernel adds the slabs in a fixed order (also into an existing gradient).
// Rows per wave and the wave count depend on the shape only: bitwise reproducible.  Roofline: MFMA for 3 channels (0.07 ms at B = 70), HBM for 1 (0.03 ms).
// --------------------------------------------------------------------------- //
struct Stem3Args {
    const float* d; const float* g; float* slab;
    int64_t d_sn, g_sn;
    int32_t d_sc, d_sd, d_sh;      // dy element strides (unit column stride)
    int32_t g_sc, g_sd, g_sh;      // x element strides (unit column stride)
    int32_t OD, OH, H;             // output depth planes, output rows, input rows
    int32_t rows, rpw;             // N * OD * OH output rows; rows per workgroup (= wave)
    int32_t J;
};

// NS stages: the DMAs of output row r + NS - 1 are issued at the top of row r (the loop is bound by the DMA round trip, not by the MFMAs, when a row's MFMAs are few)
template <int CIN, int NS>
__global__ __launch_bounds__(64) void stem3d_wgrad_kernel(const Stem3Args a) {
    constexpr int NT = 2 * CIN;                      // 32-column tiles
    constexpr int XW = CIN * 4 * 256, BUF = XW + 1024;      // words of a stage: [plane][4 rows][64 columns], then the dy row [32 oc][32 positions]
    constexpr int PER_ROW = 4 * CIN + 4;             // DMA instructions of a row
    static_assert(NS >= 2 && NS <= 4, "2-4 stages");
    __shared__ __attribute__((aligned(16))) float smem[NS * BUF];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
    const int r0 = (int)blockIdx.x * a.rpw, r1 = min(a.rows, r0 + a.rpw);
    // DMA offsets (bytes inside the sample).  dy, instruction t: lane = (channel 8 t + (lane >> 3), slot granule lane & 7) fetches granule slot ^ ((oc >> 1) & 7)
    uint32_t dvo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int oc = 8 * t + (lane >> 3);
        dvo[t] = (uint32_t)(4 * (oc * a.d_sc + 4 * ((lane & 7) ^ ((oc >> 1) & 7))));
    }
    // x, one instruction per plane: lane = (row lane >> 4, slot granule lane & 15) fetches granule slot ^ (row + 4 (plane & 1)); the row offset joins per output row
    const int xrow = lane >> 4;
    const uint32_t xg0 = (uint32_t)(16 * ((lane & 15) ^ xrow)), xg1 = (uint32_t)(16 * ((lane & 15) ^ (xrow + 4)));
    // fragment addresses (bytes inside a stage)
    const uint32_t A0 = (uint32_t)(4 * (XW + l31 * 32)), f4 = (uint32_t)(((l31 >> 1) & 7) * 16);
    // B: lane = tap (kdl = l31 >> 4, kh, kw) at position 2 st + lhi: column C = 4 st + T, T = 2 lhi + kw - 1 in [-1, 4] -> granule st + e, word cw
    const int kdl = l31 >> 4, kh = (l31 >> 2) & 3, T = 2 * lhi + (l31 & 3) - 1;
    const int e = T < 0 ? -1 : T > 3 ? 1 : 0, cw = T & 3;
    const uint32_t sw = (uint32_t)(kh + 4 * kdl);
    const uint32_t B0 = (uint32_t)(4 * (kdl * 256 + kh * 64 + cw));
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int per_n = a.OD * a.OH;

#if defined(__HIP_DEVICE_COMPILE__)
    auto issue = [&](int rr, int buf) {
        const int n = rr / per_n, rem = rr - n * per_n;
        const int od = rem / a.OH, oh = rem - od * a.OH;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g + (int64_t)n * a.g_sn), 0, 0x80000000u, 0x00020000);
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d + (int64_t)n * a.d_sn), 0, 0x80000000u, 0x00020000);
        float* xb = smem + buf * BUF;
        const int ih = 2 * oh - 1 + xrow;                            // this lane's input row
        const uint32_t rowo = (unsigned)ih < (unsigned)a.H ? (uint32_t)(4 * ih * a.g_sh) : 0x80000000u;      // outside the frame: zeros
#pragma unroll
        for (int pl = 0; pl < CIN * 4; ++pl)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(xb + pl * 256), 16, rowo + ((pl & 1) ? xg1 : xg0), 4 * ((pl >> 2) * a.g_sc + (od + (pl & 3)) * a.g_sd), 0, 0);
        const int dso = 4 * (od * a.d_sd + oh * a.d_sh);
#pragma unroll
        for (int t = 0; t < 4; ++t) __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void*)(xb + XW + t * 256), 16, dvo[t], dso, 0, 0);
    };
#else
    auto issue = [&](int, int) { (void)dvo; (void)xg0; (void)xg1; (void)per_n; (void)xrow; };
#endif

#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (r0 + i < r1) issue(r0 + i, i);
    int buf = 0;
    for (int rr = r0; rr < r1; ++rr) {
        // this row's DMAs have landed: only the younger rows' (issued after it) may still be in flight
        const int younger = min(NS - 2, r1 - 1 - rr);
        if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_ROW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_ROW) : "memory");
        if (rr + NS - 1 < r1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the fragment reads of the stage about to be refilled (the row before) have returned
            issue(rr + NS - 1, buf == 0 ? NS - 1 : buf - 1);
        }
        const char* sb = reinterpret_cast<const char*>(smem) + buf * (BUF * 4);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(sb + A0 + ((uint32_t)(16 * q) ^ f4));      // positions 4 q .. 4 q + 3 of this lane's channel
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int st = 2 * q + u;
                const float av = lhi ? a4[2 * u + 1] : a4[2 * u];
                const bool halo = (st == 0 && e < 0) || (st == 15 && e > 0);      // (false at compile time for the other steps)
                const int ge = st == 0 ? max(st + e, 0) : st == 15 ? min(st + e, 15) : st + e;
                const uint32_t bo = B0 + 16u * ((uint32_t)ge ^ sw);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float bv = *reinterpret_cast<const float*>(sb + bo + 4 * ((t >> 1) * 1024 + (t & 1) * 512));
                    if (st == 0 || st == 15) bv = halo ? 0.f : bv;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* __restrict__ out = a.slab + (int64_t)blockIdx.x * 32 * a.J;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oc = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            out[oc * a.J + 32 * t + l31] = acc[t][r];
        }
}

// --------------------------------------------------------------------------- //
// The discriminators' heads: Conv2d / Conv3d(C -> 1, 4x4(x4), stride (1,)2,2, padding (0,)1,1) on 8 x 8 planes -> 4 x 4 (discriminator.py: the last layer of the
// image / video / gradient discriminators; C = 256 / 256 / 128).  0.01-0.15 GFLOP on 8-32 MB of input: the tile kernels run them at 20-90 us per launch, latency
// bound (split-K gathers of single dwords, 128 x 128 tiles with one live column), and the iteration pays those launches one for one (profiles/r06_ab_heads.txt).
// All three passes share one lane map, chosen so that the weights are WAVE-UNIFORM: a workgroup takes four input planes (n, d); its wave q = (ph, pw) owns the
// pixels of one row / column PARITY of those planes (lane = plane, row pair, column pair).  A pixel (h, w) meets only the taps kh = 1 - ph + 2 a, kw = 1 - pw + 2 b
// (a, b in {0, 1}), i.e. 4 ND taps per channel, the same for every lane of the wave: scalar loads, 4 ND FMAs per loaded pixel, no LDS in the channel loop.
//   head_fwd_kernel   x * w into 4 ND per-lane sums over a channel range; the 16 pixels x taps that make an output position meet in LDS and are added in tap order;
//                     one partial [4 x 4] per (channel split, input plane, depth tap) -> head_fwd_combine_kernel adds splits and depth taps in order, applies act;
//   head_dgrad_kernel dx = sum over the 4 ND (depth tap, a, b) of dy[.] * w: a pure stream of the 8-32 MB gradient;
//   head_wgrad_kernel one workgroup per channel walks every plane: x * dy into 4 ND sums per lane, one wave reduction at the end, each wave stores its parity's
//                     4 ND weights (dw += for the second use of a weight in one backward).
// Every order is fixed: bitwise reproducible.  Roofline: HBM (the input read once: 4-6 us at B = 70).
// --------------------------------------------------------------------------- //
struct HeadArgs {
    const float* x; const float* w; const float* dy; float* out; float* part;
    int64_t x_sn, y_sn;
    int32_t x_sc, x_sd, y_sd;
    int32_t N, C, D, OD;           // samples, channels, input planes per sample, output planes per sample (D = OD + ND - 1)
    int32_t CS, cper;              // forward: channel splits, channels per split
    int32_t act, accumulate;
    float slope; int32_t pad;
};

// lane map shared by the three kernels: plane pj of the group, pixel (h, w) of parity (ph, pw)
#define DCV_HEAD_LANES()                                                                  \
    const int tid = threadIdx.x, lane = tid & 63;                                         \
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6), ph = q >> 1, pw = q & 1;      \
    const int pj = lane >> 4, i_ = (lane >> 2) & 3, j_ = lane & 3;                        \
    const int h = 2 * i_ + ph, w_ = 2 * j_ + pw;

template <int ND>
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadArgs a) {
    __shared__ float S[4][64][ND * 4];
    DCV_HEAD_LANES()
    const int NP = a.N * a.D;
    const int pidx = (int)blockIdx.x * 4 + pj;
    const bool valid = pidx < NP;
    const int n = valid ? pidx / a.D : 0, d = valid ? pidx - n * a.D : 0;
    const float* __restrict__ xp = a.x + (int64_t)n * a.x_sn + (int64_t)d * a.x_sd + h * 8 + w_;
    const int c0 = (int)blockIdx.y * a.cper, c1 = min(a.C, c0 + a.cper);
    float acc[ND][2][2];
#pragma unroll
    for (int kd = 0; kd < ND; ++kd) { acc[kd][0][0] = acc[kd][0][1] = acc[kd][1][0] = acc[kd][1][1] = 0.f; }
    const float* __restrict__ wq = a.w + (1 - ph) * 4 + (1 - pw);      // wave-uniform: this parity's taps are wq[kd * 16 + a * 8 + b * 2]
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const float xv = valid ? xp[(int64_t)c * a.x_sc] : 0.f;
        const float* __restrict__ wc = wq + (int64_t)c * (ND * 16);
#pragma unroll
        for (int kd = 0; kd < ND; ++kd)
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) acc[kd][aa][bb] += xv * wc[kd * 16 + aa * 8 + bb * 2];
    }
#pragma unroll
    for (int kd = 0; kd < ND; ++kd)
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) S[q][lane][kd * 4 + aa * 2 + bb] = acc[kd][aa][bb];
    __syncthreads();
    // thread -> (plane of the group, depth tap, output position): the 16 (kh, kw) contributions in tap order
    const int t_pj = tid >> 6, t_kd = (tid >> 4) & 3, oh = (tid >> 2) & 3, ow = tid & 3;
    const int t_pidx = (int)blockIdx.x * 4 + t_pj;
    if (t_kd >= ND || t_pidx >= NP) return;
    float sum = 0.f;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
        const int hh = 2 * oh - 1 + kh;
        if (hh < 0 || hh > 7) continue;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            const int ww = 2 * ow - 1 + kw;
            if (ww < 0 || ww > 7) continue;
            sum += S[(hh & 1) * 2 + (ww & 1)][t_pj * 16 + (hh >> 1) * 4 + (ww >> 1)][t_kd * 4 + (kh >> 1) * 2 + (kw >> 1)];
        }
    }
    a.part[(((int64_t)blockIdx.y * NP + t_pidx) * ND + t_kd) * 16 + oh * 4 + ow] = sum;
}

// y[n][od][oh][ow] = act(sum over channel splits, then depth taps, of part[split][n D + od + kd][kd][oh][ow])
template <int ND>
__global__ __launch_bounds__(256) void head_fwd_combine_kernel(const HeadArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot = (int64_t)a.N * a.OD * 16;
    if (i >= tot) return;
    const int pos = (int)(i & 15);
    const int64_t r = i >> 4;
    const int n = (int)(r / a.OD), od = (int)(r - (int64_t)n * a.OD);
    const int NP = a.N * a.D;
    float sum = 0.f;
    for (int cs = 0; cs < a.CS; ++cs)
#pragma unroll
        for (int kd = 0; kd < ND; ++kd) sum += a.part[(((int64_t)cs * NP + (int64_t)n * a.D + od + kd) * ND + kd) * 16 + pos];
    a.out[(int64_t)n * a.y_sn + (int64_t)od * a.y_sd + pos] = apply_act(sum, a.act, a.slope);
}

template <int ND>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const HeadArgs a) {
    DCV_HEAD_LANES()
    const int NP = a.N * a.D;
    const int pidx = (int)blockIdx.x * 4 + pj;
    if (pidx >= NP) return;
    const int n = pidx / a.D, d = pidx - n * a.D;
    // the 4 ND cotangent values this pixel meets: output plane d - kd, position (i + ph - a, j + pw - b)
    float g[ND][2][2];
#pragma unroll
    for (int kd = 0; kd < ND; ++kd)
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int od = d - kd, oh = i_ + ph - aa, ow = j_ + pw - bb;
                const bool ok = od >= 0 && od < a.OD && (unsigned)oh < 4u && (unsigned)ow < 4u;
                g[kd][aa][bb] = ok ? a.dy[(int64_t)n * a.y_sn + (int64_t)od * a.y_sd + oh * 4 + ow] : 0.f;
            }
    float* __restrict__ op = a.out + (int64_t)n * a.x_sn + (int64_t)d * a.x_sd + h * 8 + w_;
    const float* __restrict__ wq = a.w + (1 - ph) * 4 + (1 - pw);
    const int c0 = (int)blockIdx.y * a.cper, c1 = min(a.C, c0 + a.cper);
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const float* __restrict__ wc = wq + (int64_t)c * (ND * 16);
        float v = 0.f;
#pragma unroll
        for (int kd = 0; kd < ND; ++kd)
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) v += g[kd][aa][bb] * wc[kd * 16 + aa * 8 + bb * 2];
        float* o = op + (int64_t)c * a.x_sc;
        *o = a.accumulate ? *o + v : v;
    }
}

// blockIdx.x = group of CG channels, blockIdx.y = share of the plane groups; the cotangent values a pixel meets are formed once per plane and used for all CG channels.
// part[share][channel][ND * 16]: the shares are added (also into an existing gradient) by wgrad_reduce_kernel, in order.
template <int ND, int CG>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const HeadArgs a) {
    DCV_HEAD_LANES()
    const int NP = a.N * a.D, c0 = (int)blockIdx.x * CG;
    const int ngroups = (NP + 3) / 4, gper = (ngroups + (int)gridDim.y - 1) / (int)gridDim.y;
    const int g0 = (int)blockIdx.y * gper, g1 = min(ngroups, g0 + gper);
    float acc[CG][ND * 4];
#pragma unroll
    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
        for (int e = 0; e < ND * 4; ++e) acc[cc][e] = 0.f;
    // (a, b) positions of this pixel: lane constants
    int pos[4];
    bool pok[4];
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
        const int oh = i_ + ph - (ab >> 1), ow = j_ + pw - (ab & 1);
        pok[ab] = (unsigned)oh < 4u && (unsigned)ow < 4u;
        pos[ab] = pok[ab] ? oh * 4 + ow : 0;
    }
    const float* __restrict__ xc = a.x + (int64_t)c0 * a.x_sc + h * 8 + w_;
    for (int gi = g0; gi < g1; ++gi) {
        const int pidx = gi * 4 + pj;
        const bool valid = pidx < NP;
        const int n = valid ? pidx / a.D : 0, d = valid ? pidx - n * a.D : 0;
        float gv[ND * 4];
#pragma unroll
        for (int kd = 0; kd < ND; ++kd) {
            const int od = d - kd;
            const bool dok = valid && od >= 0 && od < a.OD;
            const float* __restrict__ gp = a.dy + (int64_t)n * a.y_sn + (int64_t)(dok ? od : 0) * a.y_sd;
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) gv[kd * 4 + ab] = (dok && pok[ab]) ? gp[pos[ab]] : 0.f;
        }
        const float* __restrict__ xp = xc + (int64_t)n * a.x_sn + (int64_t)d * a.x_sd;
#pragma unroll
        for (int cc = 0; cc < CG; ++cc) {
            const float xv = (valid && c0 + cc < a.C) ? xp[(int64_t)cc * a.x_sc] : 0.f;
#pragma unroll
            for (int e = 0; e < ND * 4; ++e) acc[cc][e] += xv * gv[e];
        }
    }
    // the wave's 64 lanes -> one number per (channel, kd, a, b), in a fixed butterfly order
    float* __restrict__ out = a.part + ((int64_t)blockIdx.y * a.C) * (ND * 16);
#pragma unroll
    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
        for (int e = 0; e < ND * 4; ++e) {
            float v = acc[cc][e];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0 && c0 + cc < a.C)
                out[(int64_t)(c0 + cc) * (ND * 16) + (e >> 2) * 16 + ((1 - ph) + 2 * ((e >> 1) & 1)) * 4 + (1 - pw) + 2 * (e & 1)] = v;
        }
}
#undef DCV_HEAD_LANES

// --------------------------------------------------------------------------- //
// host side: plans (index tables cached on the device per distinct geometry)
// --------------------------------------------------------------------------- //
struct DevTable {
    KEntry* dev = nullptr;      // AoS (weight packing, wgrad J rows)
    int32_t* koff = nullptr;    // SoA: byte offsets
    uint32_t* ksel = nullptr;   // SoA: tap-selection bits
    int rows = 0;
};

static std::mutex g_plan_mu;
static std::map<std::string, DevTable> g_tables;   // keys start with the device ordinal: tables live on one device

// A/B switches for tools/ (variant off when the variable is set); read once, not per call
struct Toggles {
    bool no_lds_dma, no_dstep, no_patch, no_row64, no_widen, no_widen_mfma, no_thinj_wgrad, no_wgrad_dma, no_wgrad_dma64, no_quad, no_thin_wgrad;
    int half_m;
    bool no_ragged, no_wgrad_d16;
    Toggles() {
        auto on = [](const char* n) { return getenv(n) != nullptr; };
        no_lds_dma = on("DCV_NO_LDS_DMA"); no_dstep = on("DCV_NO_DSTEP"); no_patch = on("DCV_NO_PATCH"); no_row64 = on("DCV_NO_ROW64");
        no_widen = on("DCV_NO_WIDEN"); no_widen_mfma = on("DCV_NO_WIDEN_MFMA"); no_thinj_wgrad = on("DCV_NO_THINJ_WGRAD"); no_wgrad_dma = on("DCV_NO_WGRAD_DMA"); no_wgrad_dma64 = on("DCV_NO_WGRAD_DMA64"); no_quad = on("DCV_NO_QUAD"); no_thin_wgrad = on("DCV_NO_THIN_WGRAD");
        half_m = getenv("DCV_HALF_M") ? atoi(getenv("DCV_HALF_M")) : -1;
        no_ragged = on("DCV_NO_RAGGED");
        no_wgrad_d16 = on("DCV_NO_WGRAD_D16");
    }
};
static const Toggles& toggles() {
    static const Toggles t;
    return t;
}

static std::string device_prefix() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    char b[16];
    const int n = snprintf(b, sizeof(b), "d%d|", dev);
    return std::string(b, n);
}

static int get_table(const std::string& key, const std::vector<KEntry>& host, DevTable* out) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) {
        *out = it->second;
        return DCV_OK;
    }
    DevTable t;
    t.rows = (int)host.size();
    DCV_HIP_CHECK(hipMalloc((void**)&t.dev, host.size() * sizeof(KEntry)));
    DCV_HIP_CHECK(hipMemcpy(t.dev, host.data(), host.size() * sizeof(KEntry), hipMemcpyHostToDevice));
    {
        std::vector<int32_t> off(host.size());
        std::vector<uint32_t> sel(host.size());
        for (size_t i = 0; i < host.size(); ++i) {
            off[i] = host[i].x_off * 4;
            sel[i] = host[i].tapsel;
        }
        DCV_HIP_CHECK(hipMalloc((void**)&t.koff, host.size() * sizeof(int32_t)));
        DCV_HIP_CHECK(hipMalloc((void**)&t.ksel, host.size() * sizeof(uint32_t)));
        DCV_HIP_CHECK(hipMemcpy(t.koff, off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        DCV_HIP_CHECK(hipMemcpy(t.ksel, sel.data(), sel.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    g_tables[key] = t;
    *out = t;
    return DCV_OK;
}

// One launch of the gather GEMM (one stride-parity class).
struct GatherClass {
    // per dim (d,h,w): sub-grid extent, output multiplier/offset, taps
    int o_ext[3];
    int out_mul[3], out_off[3];
    DimTaps taps[3];
    std::vector<int> tap_k[3];  // filter index along the dim for each tap u
};

struct TileCfg {
    int bn, bm;
};

static TileCfg pick_gather_tile(int OC) {
    if (OC <= 4) return {4, 64};  // thin_gather_kernel
    if (OC > 64 && OC % 128 != 0 && OC % 64 == 0 && OC <= 448) return {64, 256};   // 192, 320, 448: no padded quarter tile (ggen ngf 96)
    if (OC > 64) return {128, 128};
    if (OC > 32) return {64, 256};
    return {32, 256};
}

template <int TOC, int TM, int WOC, int WM>
static void launch_gather(const GatherArgs& a, dim3 grid, hipStream_t s) {
    if (a.structured) hipLaunchKernelGGL((gather_gemm_kernel<TOC, TM, WOC, WM, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gather_gemm_kernel<TOC, TM, WOC, WM, false>), grid, dim3(256), 0, s, a);
}

// Ragged split-K.  An op whose W workgroups (all classes, one launch) make F whole rounds of the chip's 1024 resident workgroups plus
// a small remainder E runs that remainder as an under-filled last round: the few CUs that still have a workgroup run ONE wave per
// SIMD, which cannot keep the matrix pipe busy by itself, for a whole tile's K loop (cgen.up2 at B = 70: 1120 tiles = 1 round + 96:
// 5.3 units of time against 4.375 if the work were spread evenly).  Instead the position tiles of the remainder — t1 on — are split
// over K into k parts each (blockIdx.y; ids with y > 0 are dispatched after all of y == 0, so the parts fill the chip behind the
// whole tiles), sized so that every SIMD has at least two waves: k = ceil(512 / E).  Partial sums go through the slab and
// splitk_reduce_kernel in a fixed order, as for plain split-K, so results stay bitwise reproducible.
struct RagPlan { int t1, k; };
static RagPlan rag_plan(int64_t W, int grp) {
    RagPlan p{0, 1};
    if (toggles().no_ragged || grp <= 0 || W < 1024 + 16) return p;
    const int64_t F = W / 1024;
    if (F > 2) return p;   // measured (profiles/r03_ab_ragged_splitk.txt): beyond two whole rounds the tail is too small a share to pay for the slab pass
    const int64_t t1 = (F * 1024 / grp) & ~7ll;   // whole groups of 8 position tiles (the kernel's id -> tile map deals them over the XCDs)
    const int64_t E = W - t1 * grp;
    constexpr int target = 512;   // parts to aim for (two waves per SIMD); 384 and 768 measured the same, 256 +0.6 ms per iteration
    if (t1 <= 0 || E <= 0 || E >= target) return p;
    p.t1 = (int)t1;
    p.k = (int)((target + E - 1) / E);
    if (p.k > 16) p.k = 16;
    return p;
}

// K splits for a gather launch of `blocks` workgroups over KIT 16-row steps: fill ~2 waves of
// 256 CUs x 3 blocks, keep >= 8 steps per split (thin kernel: >= 16, its 4 waves split again).
static int gather_splits(int blocks, int KIT, bool thin) {
    const int min_steps = thin ? 16 : 8;
    if (blocks >= 384 || KIT < 2 * min_steps) return 1;
    int ks = (768 + blocks - 1) / blocks;
    const int maxks = KIT / min_steps;
    if (ks > maxks) ks = maxks;
    if (ks > 64) ks = 64;
    return ks < 1 ? 1 : ks;
}

// launch the collected classes of one op as a single grid (z = class), then their split-K reduces
static int flush_packs(const float* w, const PackArgs& packs, int n, int kmax, int OC, int OCp, int64_t ws_o, hipStream_t stream) {
    const int64_t tot = (int64_t)kmax * OCp;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((tot + 255) / 256), (unsigned)n), dim3(256), 0, stream, w, packs, OC, OCp, ws_o);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

static int flush_pending(GatherArgsPack& pend, int n, dim3 grid, const TileCfg& tc, int KS, int OC, hipStream_t stream) {
#ifdef DCV_DEBUG_TIMING      // timing-experiment builds only (EXTRA_HIPCC_FLAGS=-DDCV_DEBUG_TIMING into an alt library): the shipped library cannot be told to drop its stores
    static const bool nostore = getenv("DCV_DEBUG_NOSTORE") != nullptr;
    if (nostore) for (int i = 0; i < n; ++i) pend.c[i].p_pad2 = 0x5701;
#endif
    for (int i = n; i < 4; ++i) pend.c[i] = pend.c[0];
    {   // grid.x comes in as the largest class's (oc tiles x position tiles); see the kernel's id -> tile mapping
        const unsigned tiles_oc = (unsigned)(pend.c[0].OCp / tc.bn);
        const unsigned ntm = (grid.x + tiles_oc - 1) / tiles_oc;
        grid.x = (ntm + 7) / 8 * 8 * tiles_oc * (unsigned)n;
        grid.z = 1;
        pend.c[0].pad0 = n;
    }
    const bool ds = pend.c[0].structured == 2, pt = pend.c[0].patch != 0;
    const int bfm = eff_precision();
#define DCV_LAUNCH_DMA1(A, B, C_, D, BF_)                                                                                     \
    {                                                                                                                         \
        if (ds && pt) hipLaunchKernelGGL((gather_gemm_dma_kernel<A, B, C_, D, true, true, BF_>), grid, dim3(256), 0, stream, pend); \
        else if (ds) hipLaunchKernelGGL((gather_gemm_dma_kernel<A, B, C_, D, true, false, BF_>), grid, dim3(256), 0, stream, pend); \
        else if (pt) hipLaunchKernelGGL((gather_gemm_dma_kernel<A, B, C_, D, false, true, BF_>), grid, dim3(256), 0, stream, pend); \
        else hipLaunchKernelGGL((gather_gemm_dma_kernel<A, B, C_, D, false, false, BF_>), grid, dim3(256), 0, stream, pend);        \
    }
#define DCV_LAUNCH_DMA(A, B, C_, D)                                                                                           \
    {                                                                                                                         \
        if (bfm == 2) DCV_LAUNCH_DMA1(A, B, C_, D, 2) else if (bfm == 1) DCV_LAUNCH_DMA1(A, B, C_, D, 1) else DCV_LAUNCH_DMA1(A, B, C_, D, 0) \
    }
    if (tc.bn == 128 && tc.bm == 64) DCV_LAUNCH_DMA(2, 1, 2, 2)
    else if (tc.bn == 64 && tc.bm == 128) DCV_LAUNCH_DMA(2, 1, 1, 4)
    else if (tc.bn == 128) DCV_LAUNCH_DMA(2, 2, 2, 2)
    else if (tc.bn == 64) DCV_LAUNCH_DMA(2, 2, 1, 4)
    else DCV_LAUNCH_DMA(1, 2, 1, 4)
#undef DCV_LAUNCH_DMA
#undef DCV_LAUNCH_DMA1
    DCV_NOTE_KERNEL("gather_gemm_dma_kernel<%s, %s, %s, %s> (%d x %d tile, %d class%s in one launch%s%s)",
                    tc.bn == 128 ? (tc.bm == 64 ? "2, 1, 2, 2" : "2, 2, 2, 2") : tc.bn == 64 ? (tc.bm == 128 ? "2, 1, 1, 4" : "2, 2, 1, 4") : "1, 2, 1, 4",
                    ds ? "true" : "false", pt ? "true" : "false", bfm == 2 ? "2" : bfm ? "1" : "0", tc.bn, tc.bm, n, n == 1 ? "" : "es", KS > 1 ? (pend.c[0].rag_m0 > 0 ? ", ragged split-K" : ", split-K") : "",
                    bfm == 2 ? ", f32x6: fp32 on the bf16 pipe" : bfm ? ", bf16 products" : "");
    DCV_LAUNCH_CHECK();
    if (KS > 1) {
        int64_t tot = 0;
        for (int i = 0; i < n; ++i) tot = std::max<int64_t>(tot, (int64_t)OC * (pend.c[i].slab_mp / 4));
        if (tot == 0) return DCV_OK;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256), (unsigned)n), dim3(256), 0, stream, pend, KS);
        DCV_LAUNCH_CHECK();
    }
    return DCV_OK;
}

// Which structured K walk (index-table-free, LDS-DMA kernel) a class of a gather op takes: 0 none (table-driven register staging),
// 2 depth-step order, 1 every step covers 16/T whole channels with all T taps, 3 4x4 inner taps with step = (channel, un-padded depth tap).
// Decided HERE ONLY: the ragged split-K plan, the packed-weight format and the launch setup all read this one answer.
static int structured_walk(const GatherClass& c, const dcv_dims5& xd, int RC, bool dstep, bool thin) {
    if (thin) return 0;
    if (dstep) return 2;
    const int nd = c.taps[0].n, THW = c.taps[1].n * c.taps[2].n, T = nd * THW;
    const int64_t sc4 = xd.sc * 4;
    if (T > 0 && 16 % T == 0 && RC % (16 / T) == 0 && sc4 * (16 / T) < (1ll << 30)) return 1;
    if (THW == 16 && (nd == 2 || nd == 4 || nd == 8) && c.taps[0].mul == 1 && c.taps[0].base == 0 &&
        c.taps[0].delta[nd - 1] == nd - 1 && c.o_ext[0] + nd - 1 <= c.taps[0].size && sc4 < (1ll << 30)) return 3;
    return 0;
}

// bytes per packed weight element: fp32 / bf16 packs fit 4; the three bf16 planes of the fp32-on-bf16 mode need 6
static inline size_t pack_elem_bytes() { return eff_precision() == 2 ? 6 : sizeof(float); }

// The generic driver: reduce over `RC` channels of tensor `x` (dims xd) into `OC`
// channels of tensor `y` (dims yd); classes describe position/tap relations;
// weight element (oc, rc, kd, kh, kw) lives at oc*ws_o + rc*ws_r + ((kd*KH)+kh)*KW+kw.
static int run_gather(const float* x, const dcv_dims5& xd, float* y, const dcv_dims5& yd, const float* w,
                      int RC, int OC, int64_t ws_o, int64_t ws_r, int KH, int KW,
                      const std::vector<GatherClass>& classes, int act, float slope, int accumulate,
                      void* ws, size_t ws_bytes, hipStream_t stream, const char* tag,
                      float* stat = nullptr, size_t stat_bytes = 0, int* stat_parts = nullptr, const dcv_wpack* pack = nullptr,
                      const float* gate = nullptr, float gate_slope = 0.f) {
    TileCfg tc = pick_gather_tile(OC);
    if (tc.bn >= 64) {
        // Tail regime: the op's workgroups (all classes, one launch) make a little more than a whole number of rounds of
        // the chip's 1024 resident workgroups, or do not fill one round.  Tiles of half as many positions then waste half as
        // much (a partial round of short workgroups) and fill an under-filled chip; measured on the 4x4 / 8x8-spatial
        // layers.  DCV_HALF_M = 0 / 1 forces the choice for A/B runs.
        int64_t W = 0;
        int ncls0 = 0;
        for (const GatherClass& c : classes) {
            if (c.taps[0].n * c.taps[1].n * c.taps[2].n == 0 || c.o_ext[0] <= 0 || c.o_ext[1] <= 0 || c.o_ext[2] <= 0) continue;
            const int64_t Mc = (int64_t)yd.n * c.o_ext[0] * c.o_ext[1] * c.o_ext[2];
            W += ((OC + tc.bn - 1) / tc.bn) * ((Mc + tc.bm - 1) / tc.bm);
            ++ncls0;
        }
        const double rounds = (double)W / 1024.0, frac = rounds - (double)(int64_t)rounds;
        bool half = (rounds > 0.4 && rounds < 1.0) || (rounds >= 1.0 && rounds < 3.0 && frac > 0.15 && frac < 0.55);
        if (toggles().half_m >= 0) half = toggles().half_m != 0;
        // a little more than a whole number of rounds: whole tiles with a K-split tail (below) rather than half tiles, where the tail qualifies
        // (and every class still has the >= 384 workgroups below which gather_splits splits the whole op)
        if (half && rounds >= 1.0 && toggles().half_m < 0 && !toggles().no_lds_dma && ncls0 > 0 && W / ncls0 >= 384 &&
            rag_plan(W, ((OC + tc.bn - 1) / tc.bn) * ncls0).k > 1) half = false;
        if (half) tc.bm /= 2;
    }
    // ragged split-K plan of this op (LDS-DMA MFMA path only; decided per op because the classes share one launch)
    RagPlan rag{0, 1};
    if (tc.bn >= 64 && !toggles().no_lds_dma) {
        int64_t W = 0;
        int ncls = 0;
        for (const GatherClass& c : classes) {
            if (c.taps[0].n * c.taps[1].n * c.taps[2].n == 0 || c.o_ext[0] <= 0 || c.o_ext[1] <= 0 || c.o_ext[2] <= 0) continue;
            const int64_t Mc = (int64_t)yd.n * c.o_ext[0] * c.o_ext[1] * c.o_ext[2];
            W += ((OC + tc.bn - 1) / tc.bn) * ((Mc + tc.bm - 1) / tc.bm);
            ++ncls;
        }
        rag = rag_plan(W, ((OC + tc.bn - 1) / tc.bn) * ncls);
        // the fused BatchNorm partial sums come from the direct epilogue only: a split op makes the BN op read y once more,
        // which is not worth it for the large activations (their tail is a few per cent of many rounds anyway)
        if (stat != nullptr && (int64_t)OC * yd.n * yd.d * yd.h * yd.w * 4 > (160ll << 20)) rag.k = 1;
    }
    if (gate && tc.bn == 4) return fail(DCV_EUNSUPPORTED, "%s: the gated epilogue is not built into the thin (OC <= 4) kernels", tag);
    if (gate)   // a stride-parity class without taps (stride > kernel) has positions no workgroup visits: they would stay un-gated
        for (const GatherClass& c : classes)
            if (c.taps[0].n * c.taps[1].n * c.taps[2].n == 0 && c.o_ext[0] > 0 && c.o_ext[1] > 0 && c.o_ext[2] > 0)
                return fail(DCV_EUNSUPPORTED, "%s: gated epilogue with a tap-less position class", tag);
    // packed weights: in the caller's buffer when one is given (and already valid when pack->ready), else in `ws`
    char* const pk_base = pack && pack->buf ? reinterpret_cast<char*>(pack->buf) : nullptr;
    const bool pk_ready = pk_base && pack->ready;
    size_t pk_off = 0;
    const int OCp = (OC + tc.bn - 1) / tc.bn * tc.bn;
    // fused BatchNorm partial sums (forward only): one row of {sum, sum^2} per (class, position tile)
    int stat_ntm = 0, stat_ncls = 0, stat_ci = 0;
    bool stat_ok = stat != nullptr && act == DCV_ACT_NONE && !accumulate && tc.bn != 4;
    if (stat_ok) {
        int ntm_min = INT32_MAX;
        for (const GatherClass& c : classes) {
            if (c.taps[0].n * c.taps[1].n * c.taps[2].n == 0 || c.o_ext[0] <= 0 || c.o_ext[1] <= 0 || c.o_ext[2] <= 0) continue;
            const int64_t Mc = (int64_t)yd.n * c.o_ext[0] * c.o_ext[1] * c.o_ext[2];
            const int ntm = (int)((Mc + tc.bm - 1) / tc.bm);
            stat_ntm = std::max<int>(stat_ntm, ntm);
            ntm_min = std::min<int>(ntm_min, ntm);
            ++stat_ncls;
        }
        const size_t need = (size_t)stat_ncls * stat_ntm * OCp * 2 * sizeof(float);
        if (need == 0 || need > stat_bytes) stat_ok = false;
        // every (class, position tile) row is written by exactly one workgroup when the classes have the same tile
        // count (the usual case); only ragged classes leave rows that must read as zero
        else if (ntm_min != stat_ntm) DCV_HIP_CHECK(hipMemsetAsync(stat, 0, need, stream));
    }
    if (stat_parts) *stat_parts = 0;
    size_t ws_off = 0;
    GatherArgsPack pend;
    PackArgs packs;
    memset(&packs, 0, sizeof(packs));
    int npack = 0, packmax = 0;
    int npend = 0, KSpend = 1, OCpend = 0;
    dim3 pend_grid(0, 1, 1);
    for (const GatherClass& c : classes) {
        const int T = c.taps[0].n * c.taps[1].n * c.taps[2].n;
        if (T == 0 || c.o_ext[0] <= 0 || c.o_ext[1] <= 0 || c.o_ext[2] <= 0) continue;
        const int64_t K = (int64_t)RC * T;
        const int KIT = (int)((K + 15) / 16);
        const int64_t M64 = (int64_t)yd.n * c.o_ext[0] * c.o_ext[1] * c.o_ext[2];
        if (M64 >= (1ll << 31)) return fail(DCV_EUNSUPPORTED, "%s: too many positions", tag);
        // ---- depth-step K order?  (scatter form, stride-1 depth with taps that leave the tensor) ----
        bool dstep = false;
        {
            const DimTaps& t0 = c.taps[0];
            const int nd = t0.n;
            if (tc.bn != 4 && (nd == 2 || nd == 4 || nd == 8) && c.taps[1].n * c.taps[2].n == 4 && RC % 4 == 0 && t0.mul == 1 &&
                xd.sc * 16 < (1ll << 30) && xd.sd * 4 * nd < (1ll << 30) && !toggles().no_lds_dma && !toggles().no_dstep) {
                bool consecutive = true;
                for (int u = 0; u < nd; ++u) consecutive = consecutive && t0.delta[u] == -u;
                const bool padded = t0.base - (nd - 1) < 0 || c.o_ext[0] - 1 + t0.base >= t0.size;
                dstep = consecutive && padded;
            }
        }
        // ---- K table (cached) ----
        std::string key = device_prefix() + tag;
        {
            char buf[512];
            int nn = snprintf(buf, sizeof(buf), "|g%d|%d|%d|%lld|%lld|%d|%d|%lld|%lld|%lld|%lld|", (int)dstep, RC, OC, (long long)ws_o, (long long)ws_r, KH, KW,
                              (long long)xd.sc, (long long)xd.sd, (long long)xd.sh, (long long)xd.sw);
            key.append(buf, nn);
            for (int d = 0; d < 3; ++d) {
                nn = snprintf(buf, sizeof(buf), "%d:", c.taps[d].n);
                key.append(buf, nn);
                for (int u = 0; u < c.taps[d].n; ++u) {
                    nn = snprintf(buf, sizeof(buf), "%d,%d;", c.taps[d].delta[u], c.tap_k[d][u]);
                    key.append(buf, nn);
                }
            }
        }
        DevTable tab;
        {
            std::lock_guard<std::mutex> lk(g_plan_mu);
            auto it = g_tables.find(key);
            if (it != g_tables.end()) tab = it->second;
        }
        if (!tab.dev) {
            std::vector<KEntry> host((size_t)KIT * 16);
            size_t k = 0;
            // K order (rc, ud, uh, uw); depth-step order: (rc / 4, ud descending, rc % 4, uh, uw)
            const int ND = c.taps[0].n, NI = c.taps[1].n * c.taps[2].n;
            for (int64_t kk = 0; kk < K; ++kk) {
                int rc, ud, ui;
                if (dstep) {
                    ui = (int)(kk % NI);
                    const int rcl = (int)(kk / NI % 4);
                    ud = ND - 1 - (int)(kk / (NI * 4) % ND);
                    rc = (int)(kk / ((int64_t)NI * 4 * ND)) * 4 + rcl;
                } else {
                    ui = (int)(kk % NI);
                    ud = (int)(kk / NI % ND);
                    rc = (int)(kk / ((int64_t)NI * ND));
                }
                const int uh = ui / c.taps[2].n, uw = ui % c.taps[2].n;
                        {
                            KEntry e;
                            const int64_t xo = (int64_t)rc * xd.sc + (int64_t)c.taps[0].delta[ud] * xd.sd +
                                               (int64_t)c.taps[1].delta[uh] * xd.sh + (int64_t)c.taps[2].delta[uw] * xd.sw;
                            if (xo > INT32_MAX / 4 || xo < INT32_MIN / 4) return fail(DCV_EUNSUPPORTED, "%s: tensor too large for 32-bit offsets", tag);
                            e.x_off = (int32_t)xo;
                            e.tapsel = (1u << ud) | (1u << (8 + uh)) | (1u << (16 + uw));
                            e.w_off = (int32_t)(rc * ws_r + ((int64_t)c.tap_k[0][ud] * KH + c.tap_k[1][uh]) * KW + c.tap_k[2][uw]);
                            e.pad = 0;
                            host[k++] = e;
                        }
            }
            for (; k < host.size(); ++k) host[k] = KEntry{0, 1u << 31, 0, 0};
            int rc_ = get_table(key, host, &tab);
            if (rc_ != DCV_OK) return rc_;
        }
        // ---- split-K decision: under-filled grids with a long K loop ----
        const int Mp = (int)((M64 + tc.bm - 1) / tc.bm * tc.bm);
        const int blocks = (OCp / tc.bn) * (Mp / tc.bm);
        // the classes of an op share one launch, so the grid to fill is all of them together (depth-step classes
        // skip a varying share of their steps and measured better with the per-class count)
        int KS = gather_splits(tc.bn != 4 && !dstep ? blocks * (int)classes.size() : blocks, KIT, tc.bn == 4);
        if (t_bnview) KS = 1;      // the normalise-on-load operand exists for the row kernel only (small batches would otherwise take the split-K gather)
        // ragged split-K (rag_plan): this class will take the LDS-DMA kernel (same conditions as the structured-walk choice below),
        // the op is not split as a whole, and a part keeps at least 8 K steps
        int rag_m0 = 0;
        const int walk = structured_walk(c, xd, RC, dstep, tc.bn == 4);
        const bool will_dma = walk != 0 && !toggles().no_lds_dma;   // (the toggle keeps the structured walk for the register-staged kernel)
        if (KS == 1 && rag.k > 1 && will_dma && KIT / rag.k >= 8) {
            KS = rag.k;
            rag_m0 = (int)std::min<int64_t>((int64_t)rag.t1 * tc.bm, Mp);
        } else if (rag.k > 1 && KS == 1) {
            // this class cannot take the ragged split: the classes after it do not either.  Classes already planned keep theirs — the
            // pending-launch logic below flushes whenever grid.y differs, so the two groups go out as separate launches.
            rag.k = 1;
        }
        const int kper = (KIT + KS - 1) / KS;
        const int KS2 = (KIT + kper - 1) / kper;
        const int slab_mp = Mp - rag_m0;
        // ---- pack weights ----
        const size_t wp_bytes = align_up((size_t)KIT * 16 * OCp * pack_elem_bytes(), 256);
        const size_t slab_bytes = KS2 > 1 ? align_up(std::max<size_t>((size_t)KS2 * OCp * slab_mp * sizeof(float), 256), 256) : 0;
        float* wp;
        if (pk_base) {
            if (pk_off + wp_bytes > pack->bytes) return fail(DCV_EWORKSPACE, "%s: packed-weight buffer too small (%zu needed, %zu given)", tag, pk_off + wp_bytes, pack->bytes);
            wp = reinterpret_cast<float*>(pk_base + pk_off);
            pk_off += wp_bytes;
        } else {
            if (ws_off + wp_bytes > ws_bytes) return fail(DCV_EWORKSPACE, "%s: workspace too small (%zu needed, %zu given)", tag, ws_off + wp_bytes, ws_bytes);
            wp = reinterpret_cast<float*>(static_cast<char*>(ws) + ws_off);
            ws_off += wp_bytes;
        }
        if (ws_off + slab_bytes > ws_bytes) return fail(DCV_EWORKSPACE, "%s: workspace too small (%zu needed, %zu given)", tag, ws_off + slab_bytes, ws_bytes);
        float* slab = KS2 > 1 ? reinterpret_cast<float*>(static_cast<char*>(ws) + ws_off) : nullptr;
        ws_off += slab_bytes;
        if (!pk_ready) {
            packs.ktab[npack] = tab.dev;
            packs.wp[npack] = wp;
            packs.K16[npack] = KIT * 16;
            packs.fmt[npack] = will_dma ? eff_precision() : 0;   // 0 fp32 [k][OCp], 1 bf16, 2 three bf16 planes: the LDS-DMA kernel instance of that precision reads it
            if (KIT * 16 > packmax) packmax = KIT * 16;
            ++npack;
        }
        if (npack == 4) {
            int rcp = flush_packs(w, packs, npack, packmax, OC, OCp, ws_o, stream);
            if (rcp != DCV_OK) return rcp;
            npack = 0;
            packmax = 0;
        }
        // ---- GEMM ----
        GatherArgs a;
        memset(&a, 0, sizeof(a));
        a.x = x;
        a.y = y;
        a.wp = wp;
        a.koff = tab.koff;
        a.ksel = tab.ksel;
        a.M = (int)M64;
        a.OC = OC;
        a.OCp = OCp;
        a.KIT = KIT;
        a.OD = c.o_ext[0];
        a.OH = c.o_ext[1];
        a.OW = c.o_ext[2];
        a.div_sp = make_fastdiv((uint32_t)(c.o_ext[0] * c.o_ext[1] * c.o_ext[2]));
        a.div_hw = make_fastdiv((uint32_t)(c.o_ext[1] * c.o_ext[2]));
        a.div_w = make_fastdiv((uint32_t)c.o_ext[2]);
        a.td = c.taps[0];
        a.th = c.taps[1];
        a.tw = c.taps[2];
        a.x_sn = xd.sn;
        // per-block 32-bit offset budget: samples touched by one M tile
        {
            const int64_t per = (int64_t)c.o_ext[0] * c.o_ext[1] * c.o_ext[2];
            const int64_t span = (tc.bm / per + 2) * (xd.sn < 0 ? -xd.sn : xd.sn) + (int64_t)RC * (xd.sc < 0 ? -xd.sc : xd.sc);
            if (span >= (1ll << 29) || xd.sd > INT32_MAX / 256 || xd.sh > INT32_MAX / 256 || xd.sw > INT32_MAX / 256)
                return fail(DCV_EUNSUPPORTED, "%s: input too large for 32-bit block offsets", tag);
            // the epilogue's buffer stores: sample-relative byte offsets below 2^31, channel offsets below 2^32
            const int64_t yspan = (tc.bm / per + 2) * yd.sn + (int64_t)yd.d * yd.sd + (int64_t)yd.h * yd.sh + (int64_t)yd.w * yd.sw + 8 * yd.sc;
            if (yd.sn < 0 || yd.sc < 0 || yd.sd < 0 || yd.sh < 0 || yd.sw < 0 || yspan >= (1ll << 29) || (int64_t)OCp * yd.sc >= (1ll << 30))
                return fail(DCV_EUNSUPPORTED, "%s: output too large for 32-bit block offsets", tag);
        }
        a.x_sd = (int32_t)xd.sd;
        a.x_sh = (int32_t)xd.sh;
        a.x_sw = (int32_t)xd.sw;
        a.y_sn = yd.sn;
        a.y_sc = yd.sc;
        a.y_sd = yd.sd * c.out_mul[0];
        a.y_sh = yd.sh * c.out_mul[1];
        a.y_sw = yd.sw * c.out_mul[2];
        a.y_off = yd.sd * c.out_off[0] + yd.sh * c.out_off[1] + yd.sw * c.out_off[2];
        a.act = act;
        a.slope = slope;
        a.accumulate = accumulate;
        a.slab = slab;
        a.kper = kper;
        a.Mp = Mp;
        a.slab_mp = slab_mp;
        a.rag_m0 = rag_m0;
        a.gate = gate;
        a.gate_slope = gate_slope;
        a.stat = stat_ok && KS2 == 1 ? stat : nullptr;
        a.stat_ntm = stat_ntm;
        a.stat_cls = stat_ci++;
        bool thin_struct = false;
        if (tc.bn == 4 && (T == 4 || T == 9 || T == 16) && xd.sc * 4 < (1ll << 30) && (KS2 == 1 || (kper * 16) % T == 0)) {
            // K order is (rc, ud, uh, uw): tap t of every channel has the same relative offset
            thin_struct = true;
            const int nh = c.taps[1].n, nw = c.taps[2].n, THW = nh * nw;
            a.s_stepA = (int32_t)(xd.sc * 4);
            for (int t = 0; t < T; ++t) {
                const int ud = t / THW, uh = (t / nw) % nh, uw = t % nw;
                a.s_local[t] = (int32_t)(4 * (c.taps[0].delta[ud] * xd.sd + c.taps[1].delta[uh] * xd.sh + c.taps[2].delta[uw] * xd.sw));
                a.s_sel[t] = (1u << ud) | (1u << (8 + uh)) | (1u << (16 + uw));
            }
        }
        // ---- structured K walk? (K order is (rc, ud, uh, uw), 16 rows per step) ----
        {
            const int nd = c.taps[0].n, nh = c.taps[1].n, nw = c.taps[2].n, THW = nh * nw;
            const int64_t sc4 = xd.sc * 4;
            a.structured = 0;
            if (walk == 2) {
                // step = (4 channels, one depth tap, 2x2 inner taps); depth taps walk upwards from the farthest one
                a.structured = 2;
                a.s_log2p = nd == 2 ? 1 : nd == 4 ? 2 : 3;
                a.s_stepA = (int32_t)(sc4 * 4);
                a.s_stepD = (int32_t)(4 * xd.sd);
                a.x_back = (int32_t)((nd - 1) * xd.sd);
                for (int r = 0; r < 16; ++r) {
                    const int rcl = r / 4, uh = (r % 4) / nw, uw = r % nw;
                    a.s_local[r] = (int32_t)(4 * (rcl * xd.sc + c.taps[1].delta[uh] * xd.sh + c.taps[2].delta[uw] * xd.sw));
                    a.s_sel[r] = (1u << (8 + uh)) | (1u << (16 + uw));
                }
            } else if (walk == 1) {
                // every step covers 16/T whole channels with all T taps
                a.structured = 1;
                a.s_log2p = 0;
                a.s_stepA = (int32_t)(sc4 * (16 / T));
                a.s_stepD = 0;
                for (int r = 0; r < 16; ++r) {
                    const int rcl = r / T, t = r % T, ud = t / THW, uh = (t / nw) % nh, uw = t % nw;
                    a.s_local[r] = (int32_t)(4 * (rcl * xd.sc + c.taps[0].delta[ud] * xd.sd + c.taps[1].delta[uh] * xd.sh + c.taps[2].delta[uw] * xd.sw));
                    a.s_sel[r] = (1u << ud) | (1u << (8 + uh)) | (1u << (16 + uw));
                }
            } else if (walk == 3) {
                // 4x4 inner taps, step = (channel, depth tap); depth taps never leave the tensor (no depth padding)
                a.structured = 1;
                a.s_log2p = nd == 2 ? 1 : nd == 4 ? 2 : 3;
                a.s_stepA = (int32_t)sc4;
                a.s_stepD = (int32_t)(4 * xd.sd);
                for (int r = 0; r < 16; ++r) {
                    const int uh = r / nw, uw = r % nw;
                    a.s_local[r] = (int32_t)(4 * (c.taps[1].delta[uh] * xd.sh + c.taps[2].delta[uw] * xd.sw));
                    a.s_sel[r] = (1u << (8 + uh)) | (1u << (16 + uw));
                }
            }
        }
        // ---- patch staging? (w-contiguous operand, whole output rows per tile, aligned 16-byte granules) ----
        if (a.structured && tc.bn != 4 && !toggles().no_patch) {
            const int nd = c.taps[0].n, nh = c.taps[1].n, nw = c.taps[2].n;
            int CH = 0;   // channels per K step; a step holds ONE depth tap
            if (a.structured == 2) CH = 4;
            else if (a.s_log2p > 0 || (nd == 1 && nh * nw == 16)) CH = 1;
            else if (nd == 1 && 16 % (nh * nw) == 0) CH = 16 / (nh * nw);
            const int OW = c.o_ext[2], IW = c.taps[2].size;
            auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
            auto span = [](const DimTaps& t, int* dmin) {   // taps cover a contiguous range of n offsets?
                int lo = t.delta[0], hi = t.delta[0];
                for (int u = 1; u < t.n; ++u) { lo = std::min(lo, t.delta[u]); hi = std::max(hi, t.delta[u]); }
                *dmin = lo;
                return hi - lo + 1 == t.n;
            };
            int dminh = 0, dminw = 0;
            const bool contig = span(c.taps[1], &dminh) && span(c.taps[2], &dminw);
            const int dw1 = nw >= 2 ? c.taps[2].delta[1] - c.taps[2].delta[0] : 0;
            const bool pairs = (nw == 2 || nw == 4) && (dw1 == 1 || dw1 == -1) && (nw == 2 || c.taps[2].delta[3] - c.taps[2].delta[2] == dw1);
            const int iwmin = c.taps[2].base + dminw;
            const int IWp = IW + 8, GPR = IW / 4 + 2;
            const int l2nh = ilog2(nh), l2ow = ilog2(OW);
            const bool aligned = xd.sw == 1 && IW % 4 == 0 && xd.sh % 4 == 0 && xd.sc % 4 == 0 && xd.sd % 4 == 0 && xd.sn % 4 == 0 &&
                                 (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (a.structured != 2 || a.x_back % 4 == 0);
            if (CH > 0 && CH * nh * nw == 16 && contig && pairs && l2nh >= 0 && l2ow >= 0 && OW <= tc.bm && aligned && iwmin >= -4 &&
                (OW - 1) * c.taps[2].mul + iwmin + nw - 1 <= IW + 3 && xd.sc * 4 * CH < (1ll << 30)) {
                const int TR = tc.bm / OW;
                const int G = CH * TR * nh * GPR;
                if (G <= 4 * tc.bm) {
                    a.patch = 1;
                    a.p_G = G;
                    a.p_log2nh = l2nh;
                    a.p_log2tr = ilog2(TR);
                    a.p_log2ow = l2ow;
                    a.p_iwp = IWp;
                    a.p_ihmin = c.taps[1].base + dminh;
                    a.p_iwmin4 = iwmin + 4;
                    a.p_dw1 = dw1;
                    a.p_sc4 = (int32_t)(xd.sc * 4);
                    a.p_gpr = make_fastdiv((uint32_t)GPR);
                    for (int r = 0; r < 16; ++r) {
                        const int t = r % (nh * nw), chr = r / (nh * nw), uh = t / nw, uw = t % nw;
                        a.s_local[r] = 4 * (chr * (TR * nh * IWp) + (c.taps[1].delta[uh] - dminh) * IWp + (c.taps[2].delta[uw] - dminw));
                    }
                }
            }
        }
        const dim3 grid((unsigned)blocks, (unsigned)KS2);
        const bool dma = a.structured && tc.bn != 4 && !toggles().no_lds_dma;
        if (!dma || KS2 > 1) stat_ok = false;   // only the LDS-DMA kernel's direct epilogue produces the sums
        if (will_dma != dma) return fail(DCV_EINVAL, "%s: internal: structured_walk and the launch setup disagree", tag);
        if (!dma && npack > 0) {   // an immediate launch needs its packed weights now
            int rcp = flush_packs(w, packs, npack, packmax, OC, OCp, ws_o, stream);
            if (rcp != DCV_OK) return rcp;
            npack = 0;
            packmax = 0;
        }
        if (dma) {   // deferred: merged with the other classes of this op below
            if (npend > 0 && (pend_grid.y != grid.y || pend.c[0].KIT != a.KIT || pend.c[0].patch != a.patch)) {
                if (npack > 0) { int rcp = flush_packs(w, packs, npack, packmax, OC, OCp, ws_o, stream); if (rcp != DCV_OK) return rcp; npack = 0; packmax = 0; }
                int rc2 = flush_pending(pend, npend, pend_grid, tc, KSpend, OCpend, stream);
                if (rc2 != DCV_OK) return rc2;
                npend = 0;
            }
            pend.c[npend++] = a;
            if (npend == 1 || grid.x > pend_grid.x) pend_grid.x = grid.x;
            pend_grid.y = grid.y;
            KSpend = KS2;
            OCpend = OC;
            if (npend == 4) {
                if (npack > 0) { int rcp = flush_packs(w, packs, npack, packmax, OC, OCp, ws_o, stream); if (rcp != DCV_OK) return rcp; npack = 0; packmax = 0; }
                int rc2 = flush_pending(pend, npend, pend_grid, tc, KSpend, OCpend, stream);
                if (rc2 != DCV_OK) return rc2;
                npend = 0;
            }
            continue;
        }
        // OC <= 4, unit stride, 32- or 64-wide rows, (3x3 | 2x2 | 2x2x4) taps: the row-reuse kernel
        int rows_kind = 0;   // 1: 3x3 @64   2: 2x2 @32   3: 2x2x4 @32
        int rows_iw0 = 0;
        if (tc.bn == 4 && KS2 == 1 && xd.sw == 1 && c.taps[0].mul == 1 && c.taps[1].mul == 1 && c.taps[2].mul == 1 &&
            c.o_ext[2] == c.taps[2].size && !toggles().no_row64 && xd.sc * 4 < (1ll << 30) && xd.sd * 4 * 8 < (1ll << 30)) {
            const int nd = c.taps[0].n, nh = c.taps[1].n, nw = c.taps[2].n, OWc = c.o_ext[2], OHc = c.o_ext[1];
            if (nd == 1 && nh == 3 && nw == 3 && OWc == 64 && OHc % 4 == 0) rows_kind = 1;
            else if (nd == 1 && nh == 2 && nw == 2 && OWc == 32 && OHc % 8 == 0) rows_kind = 2;
            else if (nd == 4 && nh == 2 && nw == 2 && OWc == 32 && OHc % 8 == 0) rows_kind = 3;
            if (rows_kind) {
                auto span = [](const DimTaps& t, int* dmin) {
                    int lo = t.delta[0], hi = t.delta[0];
                    for (int u = 1; u < t.n; ++u) { lo = std::min(lo, t.delta[u]); hi = std::max(hi, t.delta[u]); }
                    *dmin = lo;
                    return hi - lo + 1 == t.n;
                };
                int dminh = 0, dminw = 0;
                const bool ok = span(c.taps[1], &dminh) && span(c.taps[2], &dminw);
                rows_iw0 = c.taps[2].base + dminw;
                if (!ok || rows_iw0 < -1 || rows_iw0 + nw - 1 > 1 || (rows_kind == 1 && rows_iw0 != -1)) rows_kind = 0;
                else {
                    a.p_ihmin = c.taps[1].base + dminh;
                    a.s_stepA = (int32_t)(xd.sc * 4);
                    for (int ud = 0; ud < nd; ++ud)
                        for (int ra = 0; ra < nh; ++ra)
                            for (int b = 0; b < nw; ++b) {
                                int uh = 0, uw = 0;
                                for (int u = 0; u < nh; ++u) if (c.taps[1].delta[u] - dminh == ra) uh = u;
                                for (int u = 0; u < nw; ++u) if (c.taps[2].delta[u] - dminw == b) uw = u;
                                a.s_local[(ud * nh + ra) * nw + b] = (ud * nh + uh) * nw + uw;
                            }
                }
            }
        }
        if (rows_kind) {
            DCV_NOTE_KERNEL("thin_rows_kernel (OC %d, kind %d)", OC, rows_kind);
            if (npack > 0) {
                int rcp = flush_packs(w, packs, npack, packmax, OC, OCp, ws_o, stream);
                if (rcp != DCV_OK) return rcp;
                npack = 0;
                packmax = 0;
            }
            const dim3 gr((unsigned)(M64 / 256));
            if (BnView* bv = t_bnview) {
                // (rows_kind 1 only: the RGB head; the operand's row pitch must be the BatchNorm input's)
                if (rows_kind != 1 || OC != 3 || bv->bx_sh != a.x_sh || bv->cbn > RC) return fail(DCV_EUNSUPPORTED, "%s: BatchNorm-on-load is built for the 3-channel 3x3 head on 64-wide rows", tag);
                DCV_NOTE_KERNEL("thin_rows_kernel (OC %d, kind %d, BatchNorm + activation of the first %d channels on load)", OC, rows_kind, bv->cbn);
                hipLaunchKernelGGL((thin_rows_kernel<3, 3, 3, 1, false, -1, true>), gr, dim3(256), 0, stream, a, RC, *bv);
                bv->used = 1;
            } else
            if (rows_kind == 1) launch_thin_rows<3, 3, 1, false, -1>(a, OC, RC, gr, stream);
            else if (rows_kind == 2 && rows_iw0 == -1) launch_thin_rows<2, 2, 1, true, -1>(a, OC, RC, gr, stream);
            else if (rows_kind == 2) launch_thin_rows<2, 2, 1, true, 0>(a, OC, RC, gr, stream);
            else if (rows_iw0 == -1) launch_thin_rows<2, 2, 4, true, -1>(a, OC, RC, gr, stream);
            else launch_thin_rows<2, 2, 4, true, 0>(a, OC, RC, gr, stream);
            DCV_LAUNCH_CHECK();
            continue;
        }
        // <= 4 gathered channels, many output channels, 3x3 / unit stride / 64-wide rows: the register-resident form
        if (tc.bn != 4 && !gate && RC <= 4 && KS2 == 1 && xd.sw == 1 && c.taps[0].n == 1 && c.taps[1].n == 3 && c.taps[2].n == 3 &&
            c.taps[1].mul == 1 && c.taps[2].mul == 1 && c.o_ext[2] == 64 && c.taps[2].size == 64 && c.o_ext[1] % 4 == 0 &&
            c.out_mul[1] == 1 && c.out_mul[2] == 1 && (size_t)RC * 9 * OCp * sizeof(float) <= 48 * 1024 && !toggles().no_widen) {
            auto span3 = [](const DimTaps& t, int* dmin) {
                int lo = std::min(t.delta[0], std::min(t.delta[1], t.delta[2])), hi = std::max(t.delta[0], std::max(t.delta[1], t.delta[2]));
                *dmin = lo;
                return hi - lo == 2;
            };
            int dminh = 0, dminw = 0;
            if (span3(c.taps[1], &dminh) && span3(c.taps[2], &dminw) && c.taps[2].base + dminw == -1) {
                a.p_ihmin = c.taps[1].base + dminh;
                a.s_stepA = (int32_t)(xd.sc * 4);
                for (int ra = 0; ra < 3; ++ra)
                    for (int b = 0; b < 3; ++b) {
                        int uh = 0, uw = 0;
                        for (int u = 0; u < 3; ++u) {
                            if (c.taps[1].delta[u] - dminh == ra) uh = u;
                            if (c.taps[2].delta[u] - dminw == b) uw = u;
                        }
                        a.s_local[ra * 3 + b] = uh * 3 + uw;
                    }
                if (npack > 0) {
                    int rcp = flush_packs(w, packs, npack, packmax, OC, OCp, ws_o, stream);
                    if (rcp != DCV_OK) return rcp;
                    npack = 0;
                    packmax = 0;
                }
                // 3 gathered channels (the RGB head's data gradient): K = 27 on the matrix pipe, bound by its stores (widen_mfma_kernel)
                if (RC == 3 && (OCp == 128 || OCp == 64) && !toggles().no_widen_mfma && xd.sc * 4 * RC < (1ll << 30)) {
                    const int OHc = c.o_ext[1];
                    const int groups = OHc % 16 == 0 ? 4 : OHc % 8 == 0 ? 2 : 1;   // rows per wave (measured 1 / 2 / 4 / 8 / 16 at B = 70: 0.57 / 0.53 / 0.53 / 0.70 / 0.80 ms)
                    const dim3 gm((unsigned)(M64 / 256 / groups));
                    if (HeadBnRequest* hb = t_headbn) {      // fused with the BatchNorm backward of the first hb->cbn channels (dcv_conv_backward_data_bn)
                        const bool ok = OCp == 128 && OC == 128 && (hb->cbn == 32 || hb->cbn == 64 || hb->cbn == 96) && !accumulate && act == DCV_ACT_NONE && c.o_ext[0] == 1 &&
                                        hb->bxd.n == yd.n && hb->bxd.c == hb->cbn && hb->bxd.d == 1 && hb->bxd.h == OHc && hb->bxd.w == 64 &&
                                        hb->bxd.sn >= 0 && hb->bxd.sc >= 0 && hb->bxd.sh >= 0 && hb->bxd.sw >= 0 &&
                                        (int64_t)hb->cbn * hb->bxd.sc + (int64_t)OHc * hb->bxd.sh + 64 * hb->bxd.sw < (1ll << 29) &&
                                        (hb->mode == 1 ? (size_t)gm.x * hb->cbn * 2 * sizeof(float) <= hb->partial_bytes
                                                       : (hb->bdxd.sn >= 0 && hb->bdxd.sc >= 0 && hb->bdxd.sh >= 0 && hb->bdxd.sw >= 0 &&
                                                          (int64_t)hb->cbn * hb->bdxd.sc + (int64_t)OHc * hb->bdxd.sh + 64 * hb->bdxd.sw < (1ll << 29)));
                        if (ok) {
                            HeadBnArgs ha;
                            memset(&ha, 0, sizeof(ha));
                            ha.g = a;
                            ha.bx = hb->bx; ha.bdx = hb->bdx; ha.gamma = hb->gamma; ha.beta = hb->beta; ha.mean = hb->mean; ha.invstd = hb->invstd;
                            ha.cst = hb->cst; ha.partial = hb->partial;
                            ha.bx_sn = hb->bxd.sn; ha.bx_sc = (int32_t)hb->bxd.sc; ha.bx_sh = (int32_t)hb->bxd.sh; ha.bx_sw = (int32_t)hb->bxd.sw;
                            ha.bdx_sn = hb->bdxd.sn; ha.bdx_sc = (int32_t)hb->bdxd.sc; ha.bdx_sh = (int32_t)hb->bdxd.sh; ha.bdx_sw = (int32_t)hb->bdxd.sw;
                            ha.cbn = hb->cbn; ha.groups = groups; ha.slope = hb->slope;
                            DCV_NOTE_KERNEL("head_bn_kernel<%d> (RGB head data gradient + BatchNorm backward of %d channels)", hb->mode, hb->cbn);
                            if (hb->mode == 1) hipLaunchKernelGGL((head_bn_kernel<1>), gm, dim3(256), 0, stream, ha);
                            else hipLaunchKernelGGL((head_bn_kernel<2>), gm, dim3(256), 0, stream, ha);
                            DCV_LAUNCH_CHECK();
                            hb->used = 1;
                            hb->nwg = (int)gm.x;
                            continue;
                        }
                    }
                    DCV_NOTE_KERNEL("widen_mfma_kernel<%d, %d>", RC, OCp / 32);
                    if (OCp == 128) hipLaunchKernelGGL((widen_mfma_kernel<3, 4>), gm, dim3(256), 0, stream, a, groups);
                    else hipLaunchKernelGGL((widen_mfma_kernel<3, 2>), gm, dim3(256), 0, stream, a, groups);
                    DCV_LAUNCH_CHECK();
                    continue;
                }
                const dim3 gw((unsigned)(M64 / 256));
                const size_t shm = (size_t)RC * 9 * OCp * sizeof(float);
                DCV_NOTE_KERNEL("widen_rows_kernel<%d>", RC);
                switch (RC) {
                    case 1: hipLaunchKernelGGL(widen_rows_kernel<1>, gw, dim3(256), shm, stream, a); break;
                    case 2: hipLaunchKernelGGL(widen_rows_kernel<2>, gw, dim3(256), shm, stream, a); break;
                    case 3: hipLaunchKernelGGL(widen_rows_kernel<3>, gw, dim3(256), shm, stream, a); break;
                    default: hipLaunchKernelGGL(widen_rows_kernel<4>, gw, dim3(256), shm, stream, a); break;
                }
                DCV_LAUNCH_CHECK();
                continue;
            }
        }
        DCV_NOTE_KERNEL("%s (%d x %d tile%s)", tc.bn == 4 ? (thin_struct ? "thin_struct_kernel" : "thin_gather_kernel") : "gather_gemm_kernel", tc.bn, tc.bm, KS2 > 1 ? ", split-K" : "");
        if (tc.bn == 4 && thin_struct) {
            const int rcps = KS2 > 1 ? kper * 16 / T : RC;   // whole channels per K split
            if (T == 4) launch_thin_struct<4>(a, OC, RC, rcps, grid, stream);
            else if (T == 9) launch_thin_struct<9>(a, OC, RC, rcps, grid, stream);
            else launch_thin_struct<16>(a, OC, RC, rcps, grid, stream);
        } else if (tc.bn == 4) hipLaunchKernelGGL(thin_gather_kernel, grid, dim3(256), 0, stream, a);
        else if (tc.bn == 128 && tc.bm == 64) launch_gather<2, 1, 2, 2>(a, grid, stream);
        else if (tc.bn == 64 && tc.bm == 128) launch_gather<2, 1, 1, 4>(a, grid, stream);
        else if (tc.bn == 128) launch_gather<2, 2, 2, 2>(a, grid, stream);
        else if (tc.bn == 64) launch_gather<2, 2, 1, 4>(a, grid, stream);
        else launch_gather<1, 2, 1, 4>(a, grid, stream);
        DCV_LAUNCH_CHECK();
        if (KS2 > 1) {
            const int64_t tot = (int64_t)OC * (Mp / 4);
            GatherArgsPack one;
            for (int i = 0; i < 4; ++i) one.c[i] = a;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256), 1), dim3(256), 0, stream, one, KS2);
            DCV_LAUNCH_CHECK();
        }
    }
    if (npack > 0) {
        int rcp = flush_packs(w, packs, npack, packmax, OC, OCp, ws_o, stream);
        if (rcp != DCV_OK) return rcp;
    }
    if (npend > 0) {
        int rc2 = flush_pending(pend, npend, pend_grid, tc, KSpend, OCpend, stream);
        if (rc2 != DCV_OK) return rc2;
    }
    if (stat_parts && stat_ok) *stat_parts = stat_ncls * stat_ntm;
    return DCV_OK;
}

static size_t gather_ws_bytes(int RC, int OC, int N, const std::vector<GatherClass>& classes) {
    const TileCfg tc = pick_gather_tile(OC);
    const int OCp = (OC + tc.bn - 1) / tc.bn * tc.bn;
    size_t tot = 0;
    for (const GatherClass& c : classes) {
        const int T = c.taps[0].n * c.taps[1].n * c.taps[2].n;
        if (T == 0 || c.o_ext[0] <= 0 || c.o_ext[1] <= 0 || c.o_ext[2] <= 0) continue;
        const int64_t K = (int64_t)RC * T;
        const int KIT = (int)((K + 15) / 16);
        tot += align_up((size_t)KIT * 16 * OCp * pack_elem_bytes(), 256);
        const int64_t M64 = (int64_t)N * c.o_ext[0] * c.o_ext[1] * c.o_ext[2];
        const int64_t Mp = (M64 + tc.bm - 1) / tc.bm * tc.bm;
        const int blocks = (int)((OCp / tc.bn) * (Mp / tc.bm));
        const int KS = gather_splits(blocks, KIT, tc.bn == 4);   // per-class count: an upper bound of run_gather's choice
        if (KS > 1) tot += align_up((size_t)KS * OCp * Mp * sizeof(float), 256);
    }
    // ragged split-K slabs (rag_plan), for either position-tile size run_gather may pick: an upper bound of its choice
    size_t rag_tot = 0;
    if (tc.bn >= 64)
        for (int bm : {tc.bm, tc.bm / 2}) {
            int64_t W = 0;
            int ncls = 0;
            for (const GatherClass& c : classes) {
                if (c.taps[0].n * c.taps[1].n * c.taps[2].n == 0 || c.o_ext[0] <= 0 || c.o_ext[1] <= 0 || c.o_ext[2] <= 0) continue;
                const int64_t Mc = (int64_t)N * c.o_ext[0] * c.o_ext[1] * c.o_ext[2];
                W += (OCp / tc.bn) * ((Mc + bm - 1) / bm);
                ++ncls;
            }
            const RagPlan rp = rag_plan(W, (OCp / tc.bn) * ncls);
            if (rp.k <= 1) continue;
            size_t t = 0;
            for (const GatherClass& c : classes) {
                if (c.taps[0].n * c.taps[1].n * c.taps[2].n == 0 || c.o_ext[0] <= 0 || c.o_ext[1] <= 0 || c.o_ext[2] <= 0) continue;
                const int64_t Mc = (int64_t)N * c.o_ext[0] * c.o_ext[1] * c.o_ext[2];
                const int64_t Mp = (Mc + bm - 1) / bm * bm;
                const int64_t m0 = std::min<int64_t>((int64_t)rp.t1 * bm, Mp);
                t += align_up((size_t)rp.k * OCp * (size_t)(Mp - m0) * sizeof(float) + 256, 256);
            }
            rag_tot = std::max(rag_tot, t);
        }
    return tot + rag_tot + 256;
}

static size_t gather_pack_bytes(int RC, int OC, const std::vector<GatherClass>& classes) {
    const TileCfg tc = pick_gather_tile(OC);
    const int OCp = (OC + tc.bn - 1) / tc.bn * tc.bn;
    size_t tot = 0;
    for (const GatherClass& c : classes) {
        const int T = c.taps[0].n * c.taps[1].n * c.taps[2].n;
        if (T == 0 || c.o_ext[0] <= 0 || c.o_ext[1] <= 0 || c.o_ext[2] <= 0) continue;
        const int KIT = (int)(((int64_t)RC * T + 15) / 16);
        tot += align_up((size_t)KIT * 16 * OCp * pack_elem_bytes(), 256);
    }
    return tot;
}

// "direct" relation: gathered position = o*stride - pad + k  (conv fprop, convT dgrad)
static std::vector<GatherClass> direct_classes(const int k[3], const int s[3], const int p[3], const int o_ext[3], const int in_ext[3]) {
    GatherClass c;
    for (int d = 0; d < 3; ++d) {
        c.o_ext[d] = o_ext[d];
        c.out_mul[d] = 1;
        c.out_off[d] = 0;
        DimTaps& t = c.taps[d];
        memset(&t, 0, sizeof(t));
        t.n = k[d];
        t.mul = s[d];
        t.base = -p[d];
        t.size = in_ext[d];
        for (int u = 0; u < k[d]; ++u) {
            t.delta[u] = u;
            c.tap_k[d].push_back(u);
        }
    }
    return {c};
}

// "scatter" relation: produced position i receives tap k from gathered position
// o = (i + p - k)/s when divisible (conv dgrad, convT fprop).  One class per
// residue of i mod s in every dim.
static std::vector<GatherClass> scatter_classes(const int k[3], const int s[3], const int p[3], const int out_ext[3], const int in_ext[3]) {
    std::vector<GatherClass> out;
    for (int cd = 0; cd < s[0]; ++cd)
        for (int ch = 0; ch < s[1]; ++ch)
            for (int cw = 0; cw < s[2]; ++cw) {
                const int cls[3] = {cd, ch, cw};
                GatherClass c;
                for (int d = 0; d < 3; ++d) {
                    c.o_ext[d] = (out_ext[d] - cls[d] + s[d] - 1) / s[d];
                    c.out_mul[d] = s[d];
                    c.out_off[d] = cls[d];
                    DimTaps& t = c.taps[d];
                    memset(&t, 0, sizeof(t));
                    const int k0 = (cls[d] + p[d]) % s[d];
                    const int q = (cls[d] + p[d] - k0) / s[d];
                    t.mul = 1;
                    t.base = q;
                    t.size = in_ext[d];
                    int u = 0;
                    for (int kk = k0; kk < k[d]; kk += s[d], ++u) {
                        t.delta[u] = -u;
                        c.tap_k[d].push_back(kk);
                    }
                    t.n = u;
                }
                out.push_back(c);
            }
    return out;
}

static int check_geom(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, const char* tag) {
    if (!g || !x || !y) return fail(DCV_EINVAL, "%s: null descriptor", tag);
    if (g->kd < 1 || g->kh < 1 || g->kw < 1 || g->kd > 8 || g->kh > 8 || g->kw > 8) return fail(DCV_EINVAL, "%s: filter size out of range", tag);
    if (g->sd < 1 || g->sh < 1 || g->sw < 1 || g->pd < 0 || g->ph < 0 || g->pw < 0) return fail(DCV_EINVAL, "%s: bad stride/padding", tag);
    if (x->c != g->cin || y->c != g->cout || x->n != y->n) return fail(DCV_EINVAL, "%s: channel/batch mismatch (x.c %d cin %d, y.c %d cout %d, n %d/%d)", tag, x->c, g->cin, y->c, g->cout, x->n, y->n);
    const int xi[3] = {x->d, x->h, x->w}, yo[3] = {y->d, y->h, y->w};
    const int k[3] = {g->kd, g->kh, g->kw}, s[3] = {g->sd, g->sh, g->sw}, p[3] = {g->pd, g->ph, g->pw};
    for (int d = 0; d < 3; ++d) {
        const int expect = g->transposed ? (xi[d] - 1) * s[d] - 2 * p[d] + k[d] : (xi[d] + 2 * p[d] - k[d]) / s[d] + 1;
        if (expect != yo[d] || expect < 1) return fail(DCV_EINVAL, "%s: output extent mismatch in dim %d (%d vs %d)", tag, d, yo[d], expect);
    }
    if (x->n < 1 || x->c < 1 || y->c < 1) return fail(DCV_EINVAL, "%s: empty tensor", tag);
    return DCV_OK;
}

// ---- wgrad driver ---------------------------------------------------------- //
struct WgradTile {
    int bd, bj;
};
static WgradTile pick_wgrad_tile(int DC, int J) {
    if (J <= 32) return WgradTile{128, 32};   // stems: few gathered channels x taps
    if (DC > 64 && J > 64 && DC % 128 != 0 && DC % 64 == 0 && J % 128 == 0) return WgradTile{64, 128};   // 192, 320: whole 64-row LDS-DMA tiles (ggen ngf 96)
    if (DC > 64) return (J > 64) ? WgradTile{128, 128} : WgradTile{128, 64};
    if (DC > 32) return (J > 128 && (DC != 64 || J % 128 != 0)) ? WgradTile{64, 256} : WgradTile{64, 128};   // 64 x 128: the LDS-DMA form
    return (J > 128) ? WgradTile{32, 256} : WgradTile{32, 128};
}

static int wgrad_splits(int64_t M, int tiles) {
    // aim for ~1024 blocks, at least 8 reduction steps (256 positions) per block (measured: fewer,
    // longer blocks lose more MFMA time than the slab reduce saves)
    int64_t want = (1024 + tiles - 1) / tiles;
    int64_t maxs = (M + 255) / 256;
    int64_t s = want < maxs ? want : maxs;
    if (s < 1) s = 1;
    if (s > 4096) s = 4096;
    return (int)s;
}

template <int TD, int TJ, int WD, int WJ>
static void launch_wgrad(const WgradArgs& a, int gx, int gy, hipStream_t s) {
    if (a.log2nd >= 0) hipLaunchKernelGGL((wgrad_gemm_kernel<TD, TJ, WD, WJ, true>), dim3(gx, gy), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wgrad_gemm_kernel<TD, TJ, WD, WJ, false>), dim3(gx, gy), dim3(256), 0, s, a);
}

// dense tensor D (dims dd, channels DC), gathered tensor G (dims gd, channels GC):
// R[dc][gc][kd][kh][kw] = sum_{n,pos} D[n,dc,pos] * G[n,gc,pos*s - p + k]
// Thin weight gradient (thin_wgrad3_kernel).  Returns -1 when the geometry is not its own.
static int try_thin_wgrad(const float* D, const dcv_dims5& dd, const float* G, const dcv_dims5& gd, float* R, const int k[3], const int s[3], const int p[3],
                          void* ws, size_t ws_bytes, hipStream_t stream, const char* tag, size_t* need_only) {
    if (toggles().no_thin_wgrad) return -1;
    const int DC = dd.c, GC = gd.c;
    if (GC > 2 || dd.sw != 1 || gd.sw != 1 || dd.n != gd.n) return -1;
    if (!(k[0] == 1 && k[1] == 3 && k[2] == 3 && s[0] == 1 && s[1] == 1 && s[2] == 1 && p[0] == 0 && p[1] == 1 && p[2] == 1 && dd.d == 1 && gd.d == 1 &&
          dd.w == 64 && gd.w == 64 && dd.h == gd.h && dd.h % 4 == 0))
        return -1;
    const int dcw = GC == 1 ? 8 : 4;
    if (DC % (4 * dcw) != 0) return -1;
    const int groups = DC / (4 * dcw);
    const int P = dd.n;
    const int J = GC * 9;
    int nslab = std::max(1, std::min(P, 2048 / groups));
    const int pps = (P + nslab - 1) / nslab;
    nslab = (P + pps - 1) / pps;
    const size_t need = align_up((size_t)nslab * DC * J * sizeof(float), 256);
    if (need_only) {
        *need_only = need + 256;
        return DCV_OK;
    }
    if (!D || !G || !R || !ws) return fail(DCV_EINVAL, "%s: null pointer", tag);
    if (need > ws_bytes) return fail(DCV_EWORKSPACE, "%s: workspace too small (%zu needed, %zu given)", tag, need, ws_bytes);
    ThinWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.d = D; a.g = G; a.slab = static_cast<float*>(ws);
    a.P = P; a.DC = DC; a.OH = dd.h; a.J = J; a.pps = pps;
    a.d_sn = dd.sn; a.d_sc = dd.sc; a.d_sh = dd.sh;
    a.g_sn = gd.sn; a.g_sc = gd.sc; a.g_sh = gd.sh;
    const dim3 grid((unsigned)groups, (unsigned)nslab);
    if (GC == 1) hipLaunchKernelGGL((thin_wgrad3_kernel<1, 8>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((thin_wgrad3_kernel<2, 4>), grid, dim3(256), 0, stream, a);
    DCV_NOTE_KERNEL("thin_wgrad3_kernel<%d, %d> (%d slabs)", GC, dcw, nslab);
    DCV_LAUNCH_CHECK();
    const int64_t tot = (int64_t)DC * J;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(256), 0, stream, a.slab, R, nslab, DC, J, DC, J, t_wgrad_acc);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

// Thin-J weight gradient on the matrix pipe (thinj_wgrad_kernel).  Returns -1 when the geometry is not its own.
static int try_thinj_wgrad(const float* D, const dcv_dims5& dd, const float* G, const dcv_dims5& gd, float* R, const int k[3], const int s[3], const int p[3],
                           void* ws, size_t ws_bytes, hipStream_t stream, const char* tag, size_t* need_only) {
    if (toggles().no_thinj_wgrad || eff_precision() == 1) return -1;
    const int DC = dd.c, GC = gd.c;
    if (GC != 3 || DC % 128 != 0 || dd.sw != 1 || gd.sw != 1 || dd.n != gd.n) return -1;
    if (!(k[0] == 1 && k[1] == 3 && k[2] == 3 && s[0] == 1 && s[1] == 1 && s[2] == 1 && p[0] == 0 && p[1] == 1 && p[2] == 1 && dd.d == 1 && gd.d == 1 &&
          dd.w == 64 && gd.w == 64 && dd.h == gd.h && dd.h % 2 == 0))
        return -1;
    // 16-byte loads of the dense operand; 32-bit byte offsets inside one image
    if (dd.sc % 4 != 0 || dd.sh % 4 != 0 || dd.sn % 4 != 0 || (D && (reinterpret_cast<uintptr_t>(D) & 15) != 0) || dd.sc < 0 || dd.sh < 0 ||
        (int64_t)DC * dd.sc * 4 + (int64_t)dd.h * dd.sh * 4 >= (1ll << 31) || gd.sc > INT32_MAX / 8 || gd.sh > INT32_MAX / 8 || gd.sc < 0 || gd.sh < 0)
        return -1;
    const int OH = dd.h, rpc = OH % 8 == 0 ? 8 : OH % 4 == 0 ? 4 : 2;
    const int cpi = OH / rpc;
    const int64_t nchunk64 = (int64_t)dd.n * cpi;
    if (nchunk64 >= (1 << 30)) return -1;
    const int nchunk = (int)nchunk64, dtiles = DC / 128;
    const int target = std::max(1, 768 / dtiles);                  // three workgroups per CU (the kernel's register budget) in one round
    const int cpw = (nchunk + target - 1) / target;
    const int S = (nchunk + cpw - 1) / cpw;
    const size_t need = align_up((size_t)S * DC * 32 * sizeof(float), 256);
    if (need_only) {
        *need_only = need + 256;
        return DCV_OK;
    }
    if (!D || !G || !R || !ws) return fail(DCV_EINVAL, "%s: null pointer", tag);
    if (need > ws_bytes) return fail(DCV_EWORKSPACE, "%s: workspace too small (%zu needed, %zu given)", tag, need, ws_bytes);
    ThinJArgs a;
    memset(&a, 0, sizeof(a));
    a.d = D; a.g = G; a.slab = static_cast<float*>(ws);
    a.OH = OH; a.DC = DC; a.GC = GC; a.J = GC * 9; a.rpc = rpc; a.nchunk = nchunk; a.cpw = cpw; a.cpi = cpi;
    a.d_sn = dd.sn; a.g_sn = gd.sn;
    a.d_sc = (int32_t)dd.sc; a.d_sh = (int32_t)dd.sh; a.g_sc = (int32_t)gd.sc; a.g_sh = (int32_t)gd.sh;
    if (BnView* bv = t_bnview) {
        if (bv->bx_sc % 4 != 0 || bv->bx_sh % 4 != 0 || bv->bx_sn % 4 != 0 || (reinterpret_cast<uintptr_t>(bv->bx) & 15) != 0 || bv->cbn % 32 != 0 || bv->cbn > DC ||
            (int64_t)bv->cbn * bv->bx_sc * 4 + (int64_t)OH * bv->bx_sh * 4 >= (1ll << 31))
            return fail(DCV_EUNSUPPORTED, "%s: BatchNorm-on-load needs a 16-byte aligned BatchNorm input and whole 32-channel groups", tag);
        hipLaunchKernelGGL((thinj_wgrad_kernel<3, true>), dim3((unsigned)S, (unsigned)dtiles), dim3(256), 0, stream, a, *bv);
        bv->used = 1;
        DCV_NOTE_KERNEL("thinj_wgrad_kernel<%d> (%d slabs, BatchNorm + activation of the first %d dense channels on load)", GC, S, bv->cbn);
    } else {
    hipLaunchKernelGGL((thinj_wgrad_kernel<3>), dim3((unsigned)S, (unsigned)dtiles), dim3(256), 0, stream, a);
    DCV_NOTE_KERNEL("thinj_wgrad_kernel<%d> (%d slabs)", GC, S);
    }
    DCV_LAUNCH_CHECK();
    const int J = GC * 9;
    const int64_t tot = (int64_t)DC * J;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(256), 0, stream, a.slab, R, S, DC, J, DC, 32, t_wgrad_acc);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

// the discriminators' heads (head_fwd / head_dgrad / head_wgrad kernels): x = the module's input (N, C, D, 8, 8), y = its output (N, 1, OD, 4, 4)
static bool head_shape_ok(const int k[3], const int s[3], const int p[3], bool transposed, const dcv_dims5& x, const dcv_dims5& y) {
    static const bool off = getenv("DCV_NO_HEAD_KERNELS") != nullptr;
    if (off || transposed || eff_precision() == 1) return false;
    if (!((k[0] == 1 || k[0] == 4) && k[1] == 4 && k[2] == 4 && s[0] == 1 && s[1] == 2 && s[2] == 2 && p[0] == 0 && p[1] == 1 && p[2] == 1)) return false;
    if (y.c != 1 || x.h != 8 || x.w != 8 || x.sh != 8 || x.sw != 1 || y.h != 4 || y.w != 4 || y.sh != 4 || y.sw != 1 || x.d != y.d + k[0] - 1 || y.d < 1 || x.n != y.n) return false;
    if (x.c < 1 || x.sc < 0 || x.sd < 0 || x.sn < 0 || y.sd < 0 || y.sn < 0 || x.sc >= (1ll << 30) || x.sd >= (1ll << 30) || y.sd >= (1ll << 30)) return false;
    return (int64_t)x.n * x.d < (1ll << 28);
}
static int head_splits(const dcv_dims5& x) {      // forward: channel splits so that ~1000 workgroups exist
    const int groups = (int)(((int64_t)x.n * x.d + 3) / 4);
    int cs = std::max(1, std::min(x.c / 16, (1024 + groups - 1) / groups));
    return std::min(cs, 16);
}
static size_t head_fwd_bytes(const dcv_dims5& x, int nd) { return align_up((size_t)head_splits(x) * x.n * x.d * nd * 16 * sizeof(float), 256) + 256; }
static HeadArgs head_args(const dcv_dims5& x, const dcv_dims5& y, int nd) {
    HeadArgs a;
    memset(&a, 0, sizeof(a));
    a.x_sn = x.sn; a.y_sn = y.sn; a.x_sc = (int32_t)x.sc; a.x_sd = (int32_t)x.sd; a.y_sd = (int32_t)y.sd;
    a.N = x.n; a.C = x.c; a.D = x.d; a.OD = y.d;
    (void)nd;
    return a;
}
static int run_head_fwd(const float* x, const dcv_dims5& xd, const float* w, float* y, const dcv_dims5& yd, int nd, int act, float slope, void* ws, size_t ws_bytes, hipStream_t st) {
    const size_t need = head_fwd_bytes(xd, nd);
    if (!ws || need > ws_bytes) return fail(DCV_EWORKSPACE, "conv_fwd (head): workspace too small (%zu needed, %zu given)", need, ws_bytes);
    HeadArgs a = head_args(xd, yd, nd);
    a.x = x; a.w = w; a.out = y; a.part = static_cast<float*>(ws);
    a.CS = head_splits(xd); a.cper = (xd.c + a.CS - 1) / a.CS; a.act = act; a.slope = slope;
    const dim3 grid((unsigned)(((int64_t)xd.n * xd.d + 3) / 4), (unsigned)a.CS);
    const unsigned cb = (unsigned)(((int64_t)yd.n * yd.d * 16 + 255) / 256);
    if (nd == 1) { hipLaunchKernelGGL((head_fwd_kernel<1>), grid, dim3(256), 0, st, a); hipLaunchKernelGGL((head_fwd_combine_kernel<1>), dim3(cb), dim3(256), 0, st, a); }
    else { hipLaunchKernelGGL((head_fwd_kernel<4>), grid, dim3(256), 0, st, a); hipLaunchKernelGGL((head_fwd_combine_kernel<4>), dim3(cb), dim3(256), 0, st, a); }
    DCV_NOTE_KERNEL("head_fwd_kernel<%d> (%d plane groups x %d channel splits)", nd, (int)grid.x, a.CS);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}
static int run_head_dgrad(const float* dy, const dcv_dims5& yd, const float* w, float* dx, const dcv_dims5& xd, int nd, int accumulate, hipStream_t st) {
    HeadArgs a = head_args(xd, yd, nd);
    a.dy = dy; a.w = w; a.out = dx; a.accumulate = accumulate;
    a.CS = std::max(1, std::min(xd.c / 16, 8)); a.cper = (xd.c + a.CS - 1) / a.CS;
    const dim3 grid((unsigned)(((int64_t)xd.n * xd.d + 3) / 4), (unsigned)a.CS);
    if (nd == 1) hipLaunchKernelGGL((head_dgrad_kernel<1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((head_dgrad_kernel<4>), grid, dim3(256), 0, st, a);
    DCV_NOTE_KERNEL("head_dgrad_kernel<%d> (%d plane groups x %d channel splits)", nd, (int)grid.x, a.CS);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}
static int head_wgrad_shares(const dcv_dims5& x) {      // plane shares so that (C / 8) x shares workgroups fill the chip once
    const int groups = (int)(((int64_t)x.n * x.d + 3) / 4), cgs = (x.c + 7) / 8;
    return std::max(1, std::min(std::min(groups, 32), (512 + cgs - 1) / cgs));
}
static size_t head_wgrad_bytes(const dcv_dims5& x, int nd) { return align_up((size_t)head_wgrad_shares(x) * x.c * nd * 16 * sizeof(float), 256) + 256; }
static int run_head_wgrad(const float* x, const dcv_dims5& xd, const float* dy, const dcv_dims5& yd, float* dw, int nd, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    const size_t need = head_wgrad_bytes(xd, nd);
    if (!ws || need > ws_bytes) return fail(DCV_EWORKSPACE, "conv_bwd_weight (head): workspace too small (%zu needed, %zu given)", need, ws_bytes);
    HeadArgs a = head_args(xd, yd, nd);
    a.x = x; a.dy = dy; a.part = static_cast<float*>(ws);
    const int shares = head_wgrad_shares(xd);
    const dim3 grid((unsigned)((xd.c + 7) / 8), (unsigned)shares);
    if (nd == 1) hipLaunchKernelGGL((head_wgrad_kernel<1, 8>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((head_wgrad_kernel<4, 8>), grid, dim3(256), 0, st, a);
    DCV_NOTE_KERNEL("head_wgrad_kernel<%d> (%d channel groups x %d plane shares)", nd, (int)grid.x, shares);
    DCV_LAUNCH_CHECK();
    const int J = nd * 16;
    const int64_t tot = (int64_t)xd.c * J;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(256), 0, st, a.part, dw, shares, xd.c, J, xd.c, J, accumulate);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

// the 3-D discriminators' stems (stem3d_wgrad_kernel): -1 = not this shape
static int try_stem3d_wgrad(const float* D, const dcv_dims5& dd, const float* G, const dcv_dims5& gd, float* R, const int k[3], const int s[3], const int p[3],
                            void* ws, size_t ws_bytes, hipStream_t stream, const char* tag, size_t* need_only) {
    static const bool off = getenv("DCV_NO_STEM3D_WGRAD") != nullptr;
    if (off || eff_precision() == 1) return -1;
    const int DC = dd.c, GC = gd.c;
    if (DC != 32 || GC < 1 || GC > 3 || dd.sw != 1 || gd.sw != 1 || dd.n != gd.n) return -1;
    if (!(k[0] == 4 && k[1] == 4 && k[2] == 4 && s[0] == 1 && s[1] == 2 && s[2] == 2 && p[0] == 0 && p[1] == 1 && p[2] == 1)) return -1;
    if (gd.w != 64 || dd.w != 32 || gd.h != 2 * dd.h || dd.h < 1 || gd.d != dd.d + 3 || dd.d < 1) return -1;
    // 16-byte LDS-DMA granules: rows of both operands on 16-byte addresses
    if (gd.sn % 4 || gd.sc % 4 || gd.sd % 4 || gd.sh % 4 || dd.sn % 4 || dd.sc % 4 || dd.sd % 4 || dd.sh % 4 ||
        (G && (reinterpret_cast<uintptr_t>(G) & 15) != 0) || (D && (reinterpret_cast<uintptr_t>(D) & 15) != 0))
        return -1;
    // 32-bit byte offsets inside one sample (the kernel's buffer resources start at the sample)
    if (gd.sc < 0 || gd.sd < 0 || gd.sh < 0 || dd.sc < 0 || dd.sd < 0 || dd.sh < 0) return -1;
    if ((int64_t)(GC - 1) * gd.sc + (int64_t)(gd.d - 1) * gd.sd + (int64_t)(gd.h - 1) * gd.sh + 64 >= (1ll << 28) ||
        (int64_t)31 * dd.sc + (int64_t)(dd.d - 1) * dd.sd + (int64_t)(dd.h - 1) * dd.sh + 32 >= (1ll << 28))
        return -1;
    const int64_t rows64 = (int64_t)dd.n * dd.d * dd.h;
    if (rows64 < 1 || rows64 >= (1 << 30)) return -1;
    const int rows = (int)rows64, J = 64 * GC;
    static const int ns_env = getenv("DCV_STEM3D_STAGES") ? atoi(getenv("DCV_STEM3D_STAGES")) : 0;      // A/B only
    const int NS = ns_env >= 2 && ns_env <= 4 ? ns_env : 2;      // measured at B = 70 with the wide slab reduce: 1 channel 0.059 / 0.069 / 0.067 ms with 2 / 3 / 4 stages, 3 channels 0.130 / 0.156 / 0.221
    const int stage_bytes = (GC * 1024 + 1024) * 4;
    const int cap = 256 * std::max(1, std::min(8, (159 * 1024) / (NS * stage_bytes)));      // waves the chip holds at this LDS footprint
    const int rpw = std::max(8, (rows + cap - 1) / cap);
    const int nwg = (rows + rpw - 1) / rpw;
    const size_t need = align_up((size_t)nwg * 32 * J * sizeof(float), 256);
    if (need_only) {
        *need_only = need + 256;
        return DCV_OK;
    }
    if (!D || !G || !R || !ws) return fail(DCV_EINVAL, "%s: null pointer", tag);
    if (need > ws_bytes) return fail(DCV_EWORKSPACE, "%s: workspace too small (%zu needed, %zu given)", tag, need, ws_bytes);
    Stem3Args a;
    memset(&a, 0, sizeof(a));
    a.d = D; a.g = G; a.slab = static_cast<float*>(ws);
    a.d_sn = dd.sn; a.g_sn = gd.sn;
    a.d_sc = (int32_t)dd.sc; a.d_sd = (int32_t)dd.sd; a.d_sh = (int32_t)dd.sh;
    a.g_sc = (int32_t)gd.sc; a.g_sd = (int32_t)gd.sd; a.g_sh = (int32_t)gd.sh;
    a.OD = dd.d; a.OH = dd.h; a.H = gd.h; a.rows = rows; a.rpw = rpw; a.J = J;
#define DCV_STEM3(C_, N_) hipLaunchKernelGGL((stem3d_wgrad_kernel<C_, N_>), dim3((unsigned)nwg), dim3(64), 0, stream, a)
    if (GC == 1) { if (NS == 2) DCV_STEM3(1, 2); else if (NS == 3) DCV_STEM3(1, 3); else DCV_STEM3(1, 4); }
    else if (GC == 2) { if (NS == 2) DCV_STEM3(2, 2); else if (NS == 3) DCV_STEM3(2, 3); else DCV_STEM3(2, 4); }
    else { if (NS == 2) DCV_STEM3(3, 2); else if (NS == 3) DCV_STEM3(3, 3); else DCV_STEM3(3, 4); }
#undef DCV_STEM3
    DCV_NOTE_KERNEL("stem3d_wgrad_kernel<%d, %d stages> (%d waves x %d output rows)", GC, NS, nwg, rpw);
    DCV_LAUNCH_CHECK();
    if ((reinterpret_cast<uintptr_t>(R) & 15) == 0)
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)((8 * J + 3) / 4)), dim3(256), 0, stream, a.slab, R, nwg, 8 * J, t_wgrad_acc);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((32 * J + 63) / 64)), dim3(256), 0, stream, a.slab, R, nwg, 32, J, 32, J, t_wgrad_acc);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

static int run_wgrad(const float* D, const dcv_dims5& dd, const float* G, const dcv_dims5& gd, float* R,
                     const int k[3], const int s[3], const int p[3], void* ws, size_t ws_bytes, hipStream_t stream, const char* tag,
                     size_t* need_only) {
    const int DC = dd.c, GC = gd.c;
    const int T = k[0] * k[1] * k[2];
    const int64_t J64 = (int64_t)GC * T;
    if (J64 >= (1 << 30)) return fail(DCV_EUNSUPPORTED, "%s: J too large", tag);
    size_t thin_need = 0;
    if (head_shape_ok(k, s, p, false, gd, dd) && dd.c == 1) {      // a discriminator's head (conv: dense = dy with one channel, gathered = x)
        if (need_only) { *need_only = head_wgrad_bytes(gd, k[0]); return DCV_OK; }
        if (!D || !G || !R) return fail(DCV_EINVAL, "%s: null pointer", tag);
        return run_head_wgrad(G, gd, D, dd, R, k[0], t_wgrad_acc, ws, ws_bytes, stream);
    }
    {   // the colour generator's stem: VALU kernel
        const int rc_ = try_thin_wgrad(D, dd, G, gd, R, k, s, p, ws, ws_bytes, stream, tag, need_only ? &thin_need : nullptr);
        if (rc_ != -1 && !need_only) return rc_;
    }
    {   // the colour generator's RGB head: one MFMA column of taps, the dense operand streamed once
        size_t tj_need = 0;
        const int rc_ = try_thinj_wgrad(D, dd, G, gd, R, k, s, p, ws, ws_bytes, stream, tag, need_only ? &tj_need : nullptr);
        if (rc_ != -1 && !need_only) return rc_;
        thin_need = std::max(thin_need, tj_need);
    }
    {   // the 3-D discriminators' stems: one wave per run of output rows, everything staged once
        size_t st_need = 0;
        const int rc_ = try_stem3d_wgrad(D, dd, G, gd, R, k, s, p, ws, ws_bytes, stream, tag, need_only ? &st_need : nullptr);
        if (rc_ != -1 && !need_only) return rc_;
        thin_need = std::max(thin_need, st_need);
    }
    const int J = (int)J64;
    const WgradTile tc = pick_wgrad_tile(DC, J);
    const int DCp = (DC + tc.bd - 1) / tc.bd * tc.bd;
    const int Jp = (J + tc.bj - 1) / tc.bj * tc.bj;
    const int64_t M64 = (int64_t)dd.n * dd.d * dd.h * dd.w;
    if (M64 >= (1ll << 31)) return fail(DCV_EUNSUPPORTED, "%s: too many positions", tag);
    const int tiles = (DCp / tc.bd) * (Jp / tc.bj);
    // double-buffered LDS-DMA form: 128 x 128 tile, 4x4 inner taps, un-padded depth taps, full channel tiles
    bool dma = !toggles().no_wgrad_dma && ((tc.bd == 128 && tc.bj == 128) || (tc.bd == 64 && tc.bj == 128 && !toggles().no_wgrad_dma64)) && k[1] == 4 && k[2] == 4 &&
               (k[0] == 1 || k[0] == 2 || k[0] == 4 || k[0] == 8) && (k[0] == 1 || (p[0] == 0 && s[0] == 1)) &&
               DC % tc.bd == 0 && J % 128 == 0 && gd.sc * 4 < (1ll << 30) && dd.sc * 4 < (1ll << 30);
    int S = 1;
    int64_t chunk = 0;
    if (dma) {   // one workgroup per CU: whole rounds of 256, >= 1024 positions each
        const int slots = 256, target = 512;                          // one workgroup per CU; two rounds
        S = (target + tiles - 1) / tiles;
        const int64_t maxs = (M64 + 1023) / 1024;
        if (S > maxs) S = (int)maxs;
        if (S < 1) S = 1;
        {   // tile counts that are not powers of two (ngf 96): among the slab counts near S take the one whose last round is fullest
            auto eff = [&](int s_) { const int64_t wg = (int64_t)tiles * s_; return (double)wg / (double)((wg + slots - 1) / slots * slots); };
            int best = S;
            for (int c = std::max(1, S - S / 3); c <= S + (S + 1) / 2 && c <= maxs; ++c)
                if (eff(c) > eff(best) + 0.04) best = c;
            S = best;
        }
        chunk = ((M64 + S - 1) / S + 63) / 64 * 64;
        // its padding encoding needs the gathered operand's block-relative byte offsets below 2^30
        const int64_t per = (int64_t)dd.d * dd.h * dd.w;
        if ((chunk / per + 2) * gd.sn * 4 >= (1ll << 30)) dma = false;
    }
    if (!dma) {
        S = wgrad_splits(M64, tiles);
        chunk = ((M64 + S - 1) / S + 63) / 64 * 64;
    }
    const int S2 = (int)((M64 + chunk - 1) / chunk);
    const size_t need = align_up((size_t)S2 * DCp * Jp * sizeof(float), 256);
    if (need_only) {
        *need_only = std::max(need + 256, thin_need);
        return DCV_OK;
    }
    if (need > ws_bytes) return fail(DCV_EWORKSPACE, "%s: workspace too small (%zu needed, %zu given)", tag, need, ws_bytes);

    char kb[512];
    int nn = snprintf(kb, sizeof(kb), "%s|w|%d|%d,%d,%d|%lld|%lld|%lld|%lld", tag, GC, k[0], k[1], k[2], (long long)gd.sc, (long long)gd.sd, (long long)gd.sh, (long long)gd.sw);
    std::string key = device_prefix() + std::string(kb, nn);
    DevTable tab;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        auto it = g_tables.find(key);
        if (it != g_tables.end()) tab = it->second;
    }
    if (!tab.dev) {
        std::vector<KEntry> host((size_t)J);
        size_t q = 0;
        for (int gc = 0; gc < GC; ++gc)
            for (int ud = 0; ud < k[0]; ++ud)
                for (int uh = 0; uh < k[1]; ++uh)
                    for (int uw = 0; uw < k[2]; ++uw) {
                        KEntry e;
                        const int64_t xo = (int64_t)gc * gd.sc + (int64_t)ud * gd.sd + (int64_t)uh * gd.sh + (int64_t)uw * gd.sw;
                        if (xo > INT32_MAX / 4) return fail(DCV_EUNSUPPORTED, "%s: tensor too large for 32-bit offsets", tag);
                        e.x_off = (int32_t)xo;
                        e.tapsel = (1u << ud) | (1u << (8 + uh)) | (1u << (16 + uw));
                        e.w_off = 0;
                        e.pad = 0;
                        host[q++] = e;
                    }
        int rc_ = get_table(key, host, &tab);
        if (rc_ != DCV_OK) return rc_;
    }
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dptr = D;
    a.gptr = G;
    a.slab = static_cast<float*>(ws);
    a.jtab = tab.dev;
    a.M = (int)M64;
    a.DC = DC;
    a.J = J;
    a.DCp = DCp;
    a.Jp = Jp;
    a.chunk = (int)chunk;
    a.div_sp = make_fastdiv((uint32_t)(dd.d * dd.h * dd.w));
    a.div_hw = make_fastdiv((uint32_t)(dd.h * dd.w));
    a.div_w = make_fastdiv((uint32_t)dd.w);
    const int gext[3] = {gd.d, gd.h, gd.w};
    DimTaps* ts[3] = {&a.td, &a.th, &a.tw};
    for (int d = 0; d < 3; ++d) {
        ts[d]->n = k[d];
        ts[d]->mul = s[d];
        ts[d]->base = -p[d];
        ts[d]->size = gext[d];
        for (int u = 0; u < k[d]; ++u) ts[d]->delta[u] = u;
    }
    {
        // 32-bit byte offsets relative to the sample of a block's first position
        const int64_t per = (int64_t)dd.d * dd.h * dd.w;
        const int64_t samples = chunk / per + 2;
        const int64_t dspan = samples * dd.sn + (int64_t)DCp * dd.sc, gspan = samples * gd.sn + (int64_t)GC * gd.sc;
        if (dspan >= (1ll << 29) || gspan >= (1ll << 29)) return fail(DCV_EUNSUPPORTED, "%s: tensors too large for 32-bit block offsets", tag);
    }
    a.d_sn = dd.sn; a.g_sn = gd.sn;
    a.d_sc4 = (int32_t)(dd.sc * 4); a.d_sd = (int32_t)dd.sd; a.d_sh = (int32_t)dd.sh; a.d_sw = (int32_t)dd.sw;
    a.g_sd = (int32_t)gd.sd; a.g_sh = (int32_t)gd.sh; a.g_sw = (int32_t)gd.sw;
    // regular (4x4 inner taps, depth taps never padded) form?
    a.log2nd = -1;
    if (k[1] * k[2] == 16 && (k[0] == 1 || k[0] == 2 || k[0] == 4 || k[0] == 8) && (k[0] == 1 || (p[0] == 0 && s[0] == 1)) &&
        tc.bj >= 128 && gd.sc * 4 < (1ll << 30)) {
        a.log2nd = k[0] == 1 ? 0 : k[0] == 2 ? 1 : k[0] == 4 ? 2 : 3;
        a.g_sc4 = (int32_t)(gd.sc * 4);
        a.g_sd4 = (int32_t)(gd.sd * 4);
        for (int t = 0; t < 16; ++t) {
            const int uh = t / k[2], uw = t % k[2];
            a.hw_off4[t] = (int32_t)(4 * (uh * gd.sh + uw * gd.sw));
            a.hw_sel[t] = (1u << (8 + uh)) | (1u << (16 + uw));
        }
    }
    // 16-byte staging of the dense operand (wgrad_dma_kernel<.., D16>): its rows are contiguous, 16-byte aligned runs of positions, and tiles, chunks
    // and M are whole granules of 4 positions
    // ... and only the 64-row tile takes it: its 48 row DMAs per 64 MFMAs are what bounds it (cgen.down0 1.351 -> 1.286 ms, gdis.5 0.355 -> 0.353), while
    // the MFMA-bound 128 x 128 tile measured 4-6 % SLOWER with it (the 2-way conflicts of the swizzled fragment reads; profiles/r03_ab_wgrad_d16.txt)
    const bool d16 = !toggles().no_wgrad_d16 && tc.bd == 64 && dd.sw == 1 && dd.w % 4 == 0 && dd.sh % 4 == 0 && dd.sd % 4 == 0 && dd.sn % 4 == 0 && dd.sc % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(D) % 16) == 0 && M64 % 4 == 0 && chunk % 4 == 0 && eff_precision() != 1 && !(eff_precision() == 2 && getenv("DCV_WGRAD_X6"));
    {
        const int wbf = eff_precision() == 2 && getenv("DCV_WGRAD_X6") == nullptr ? 0 : eff_precision();
        if (dma && a.log2nd >= 0) DCV_NOTE_KERNEL("wgrad_dma_kernel<%d, %d, %s> (%d x %d tile, %d slabs%s)", tc.bd == 128 ? 2 : 1, wbf, d16 ? "true" : "false", tc.bd, tc.bj, S2,
                                                  wbf == 2 ? ", f32x6: fp32 on the bf16 pipe" : wbf ? ", bf16 products" : "");
        else DCV_NOTE_KERNEL("wgrad_gemm_kernel (%d x %d tile, %d slabs)", tc.bd, tc.bj, S2);
    }
    // fp32 on the bf16 pipe (precision 2): the weight gradient keeps the native fp32 MFMA kernel unless DCV_WGRAD_X6 is set — both of its operands are
    // activations, so both are split in registers (176 vector operations per 24 MFMAs): measured 112-128 TFLOP/s against the native kernel's 120-135
    // (profiles/r04_f32x6_layers.csv); the gather kernels, whose weights arrive pre-split, gain 1.4-1.5x
    static const bool wgrad_x6 = getenv("DCV_WGRAD_X6") != nullptr;
    if (dma && a.log2nd >= 0 && eff_precision() == 2 && wgrad_x6) {
        if (tc.bd == 128) hipLaunchKernelGGL((wgrad_dma_kernel<2, 2>), dim3(tiles, S2), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((wgrad_dma_kernel<1, 2>), dim3(tiles, S2), dim3(256), 0, stream, a);
    } else if (dma && a.log2nd >= 0 && eff_precision() == 1) {
        if (tc.bd == 128) hipLaunchKernelGGL((wgrad_dma_kernel<2, 1>), dim3(tiles, S2), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((wgrad_dma_kernel<1, 1>), dim3(tiles, S2), dim3(256), 0, stream, a);
    } else if (dma && a.log2nd >= 0 && d16) {
        hipLaunchKernelGGL((wgrad_dma_kernel<1, 0, true>), dim3(tiles, S2), dim3(256), 0, stream, a);
    } else if (dma && a.log2nd >= 0) {
        if (tc.bd == 128) hipLaunchKernelGGL((wgrad_dma_kernel<2, 0>), dim3(tiles, S2), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((wgrad_dma_kernel<1, 0>), dim3(tiles, S2), dim3(256), 0, stream, a);
    }
    else if (tc.bd == 128 && tc.bj == 32) launch_wgrad<1, 1, 4, 1>(a, tiles, S2, stream);
    else if (tc.bd == 128 && tc.bj == 128) launch_wgrad<2, 2, 2, 2>(a, tiles, S2, stream);
    else if (tc.bd == 128 && tc.bj == 64) launch_wgrad<2, 1, 2, 2>(a, tiles, S2, stream);
    else if (tc.bd == 64 && tc.bj == 256) launch_wgrad<2, 2, 1, 4>(a, tiles, S2, stream);
    else if (tc.bd == 64 && tc.bj == 128) launch_wgrad<2, 1, 1, 4>(a, tiles, S2, stream);
    else if (tc.bd == 32 && tc.bj == 256) launch_wgrad<1, 2, 1, 4>(a, tiles, S2, stream);
    else launch_wgrad<1, 1, 1, 4>(a, tiles, S2, stream);
    DCV_LAUNCH_CHECK();
    const int64_t tot = (int64_t)DC * J;
    if (J % 4 == 0 && Jp % 4 == 0 && (reinterpret_cast<uintptr_t>(R) & 15) == 0)
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((tot / 4 + 63) / 64)), dim3(256), 0, stream, a.slab, R, S2, DC, J, DCp, Jp, t_wgrad_acc);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(256), 0, stream, a.slab, R, S2, DC, J, DCp, Jp, t_wgrad_acc);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

}  // namespace dcv

using namespace dcv;

extern "C" {

// Diagnostics: resident workgroups per CU (HIP occupancy API) and register/LDS use of each GEMM
// kernel instantiation, as text.  Used by tools/ and DESIGN.md; not part of the hot path.
int dcv_debug_kernel_info(char* buf, size_t n) {
    struct Item { const char* name; const void* fn; };
    const Item items[] = {
        {"gather<2,2,2,2,S>", (const void*)gather_gemm_kernel<2, 2, 2, 2, true>}, {"gather<2,2,1,4,S>", (const void*)gather_gemm_kernel<2, 2, 1, 4, true>},
        {"gather<1,2,1,4,S>", (const void*)gather_gemm_kernel<1, 2, 1, 4, true>}, {"gather<2,2,2,2,T>", (const void*)gather_gemm_kernel<2, 2, 2, 2, false>},
        {"gather_dma<2,2,2,2>", (const void*)gather_gemm_dma_kernel<2, 2, 2, 2, false, false>}, {"gather_dma<2,2,1,4>", (const void*)gather_gemm_dma_kernel<2, 2, 1, 4, false, false>},
        {"gather_dma<1,2,1,4>", (const void*)gather_gemm_dma_kernel<1, 2, 1, 4, false, false>}, {"gather_dma_patch<2,2,2,2>", (const void*)gather_gemm_dma_kernel<2, 2, 2, 2, false, true>},
        {"thin_gather", (const void*)thin_gather_kernel},
        {"wgrad_dma", (const void*)wgrad_dma_kernel<2>}, {"wgrad_dma64", (const void*)wgrad_dma_kernel<1>}, {"wgrad<2,2,2,2,R>", (const void*)wgrad_gemm_kernel<2, 2, 2, 2, true>}, {"wgrad<2,2,1,4,R>", (const void*)wgrad_gemm_kernel<2, 2, 1, 4, true>},
        {"wgrad<2,2,2,2>", (const void*)wgrad_gemm_kernel<2, 2, 2, 2, false>}, {"wgrad<2,1,2,2>", (const void*)wgrad_gemm_kernel<2, 1, 2, 2, false>},
        {"wgrad<2,2,1,4>", (const void*)wgrad_gemm_kernel<2, 2, 1, 4, false>}, {"wgrad<2,1,1,4>", (const void*)wgrad_gemm_kernel<2, 1, 1, 4, false>},
        {"wgrad<1,2,1,4>", (const void*)wgrad_gemm_kernel<1, 2, 1, 4, false>}, {"wgrad<1,1,1,4>", (const void*)wgrad_gemm_kernel<1, 1, 1, 4, false>},
    };
    size_t off = 0;
    for (const Item& it : items) {
        int blocks = -1;
        hipFuncAttributes at;
        memset(&at, 0, sizeof(at));
        hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, it.fn, 256, 0);
        hipError_t e2 = hipFuncGetAttributes(&at, it.fn);
        int w = snprintf(buf + off, off < n ? n - off : 0, "%s: blocks/CU=%d regs=%d lds=%zu scratch=%zu (%s)\n", it.name, blocks, at.numRegs,
                         (size_t)at.sharedSizeBytes, (size_t)at.localSizeBytes, (e1 == hipSuccess && e2 == hipSuccess) ? "ok" : "query failed");
        if (w < 0) break;
        off += (size_t)w;
        if (off >= n) break;
    }
    return (int)off;
}

#ifdef DCV_STAMP
int dcv_debug_read_stamps(unsigned long long* host, int nblocks) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 24 * (size_t)nblocks, 0, hipMemcpyDeviceToHost);
}
#endif

const char* dcv_last_error(void) { return g_err; }
const char* dcv_debug_last_kernel(void) { return g_last_kernel; }
int dcv_set_precision(int mode) {
    if (mode < 0 || mode > 2) return fail(DCV_EINVAL, "set_precision: 0 = fp32, 1 = bf16 MFMA products, 2 = fp32 emulated on the bf16 matrix pipe (3 x bf16 split, 6 products)");
    g_precision.store(mode);
    return DCV_OK;
}
int dcv_get_precision(void) { return g_precision.load(); }
int dcv_version(void) { return 4; }
void dcv_abi_struct_sizes(size_t out[3]) {
    if (!out) return;
    out[0] = sizeof(dcv_dims5); out[1] = sizeof(dcv_conv_geom); out[2] = sizeof(dcv_wpack);
}
int dcv_conv_effective_precision(const dcv_conv_geom* g) {
    if (g && g->mfma >= 1 && g->mfma <= 3) return g->mfma;
    return g_precision.load() + 1;
}
uint64_t dcv_launch_count(void) { return g_launches.load(); }

// A transposed convolution of a 1x1(x1) input with stride 1 and no padding (the latent layer, generator.py:61:
// ConvTranspose2d(dim_z, 8 ngf, 4, 1, 0)) is a plain matrix product y[n, (co, kd, kh, kw)] = sum_ci x[n, ci] w[ci, (co, kd, kh, kw)]:
// as a k x k scatter it would walk k^3 taps of which all but one are padding for every output position.  When y is
// contiguous per sample it is re-described as a 1x1 transposed convolution with cout * taps output channels (weights and
// gradients are the same memory either way).
static bool latent_form(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, dcv_conv_geom* g2, dcv_dims5* y2) {
    if (!g || !x || !y || !g->transposed) return false;
    const int T = g->kd * g->kh * g->kw;
    if (T <= 1 || x->d != 1 || x->h != 1 || x->w != 1 || g->sd != 1 || g->sh != 1 || g->sw != 1 || g->pd || g->ph || g->pw) return false;
    if (y->d != g->kd || y->h != g->kh || y->w != g->kw || y->c != g->cout || (int64_t)g->cout * T >= (1ll << 30)) return false;
    // (c, d, h, w) of y contiguous (stride fields of size-1 dims are normalised to 0 by the caller)
    const int64_t sw = 1, sh = y->w, sd = (int64_t)y->h * y->w, sc = (int64_t)y->d * y->h * y->w;
    if ((y->w > 1 && y->sw != sw) || (y->h > 1 && y->sh != sh) || (y->d > 1 && y->sd != sd) || (y->c > 1 && y->sc != sc)) return false;
    *g2 = *g;
    g2->kd = g2->kh = g2->kw = 1;
    g2->cout = g->cout * T;
    *y2 = *y;
    y2->c = g->cout * T; y2->d = y2->h = y2->w = 1;
    y2->sc = 1; y2->sd = y2->sh = y2->sw = 0;
    return true;
}

// OC <= 4 scatter-form ops of the 4x4 / stride-2 / pad-1 family on 32 -> 64 wide rows: thin_quad_kernel (returns false when the
// geometry is another one and the generic class-by-class path has to run)
static bool try_thin_quad(const float* src, const dcv_dims5& sd_, float* dst, const dcv_dims5& dd, const float* w, const int k[3], const int s[3], const int p[3],
                          int act, float slope, int accumulate, hipStream_t stream) {
    if (toggles().no_quad) return false;
    const int OC = dd.c, RC = sd_.c;
    if (OC > 4 || k[1] != 4 || k[2] != 4 || s[1] != 2 || s[2] != 2 || p[1] != 1 || p[2] != 1 || s[0] != 1 || p[0] != 0 || (k[0] != 1 && k[0] != 4)) return false;
    if (sd_.w != 32 || dd.w != 64 || dd.h != 2 * sd_.h || sd_.h % 4 != 0 || sd_.sw != 1 || dd.sw != 1) return false;
    if (dd.d != sd_.d + k[0] - 1) return false;
    if ((int64_t)dd.n * dd.d * (sd_.h / 4) >= (1ll << 31) || sd_.sh * (int64_t)sd_.h >= (1ll << 30)) return false;
    QuadArgs a;
    memset(&a, 0, sizeof(a));
    a.s = src; a.y = dst; a.w = w;
    a.N = dd.n; a.RC = RC; a.OC = OC; a.SD = sd_.d; a.SH = sd_.h; a.OD = dd.d;
    a.s_sn = sd_.sn; a.s_sc = sd_.sc; a.s_sd = sd_.sd; a.s_sh = sd_.sh;
    a.y_sn = dd.sn; a.y_sc = dd.sc; a.y_sd = dd.sd; a.y_sh = dd.sh;
    const int T = k[0] * 16;
    a.w_o = T; a.w_r = (int64_t)OC * T;
    a.act = act; a.slope = slope; a.accumulate = accumulate;
    const dim3 grid((unsigned)((int64_t)dd.n * dd.d * (sd_.h / 4)));
#define DCV_QUAD(NOC_)                                                                                          \
    {                                                                                                           \
        if (k[0] == 1) hipLaunchKernelGGL((thin_quad_kernel<NOC_, 1>), grid, dim3(256), 0, stream, a);          \
        else hipLaunchKernelGGL((thin_quad_kernel<NOC_, 4>), grid, dim3(256), 0, stream, a);                    \
    }
    switch (OC) {
        case 1: DCV_QUAD(1) break;
        case 2: DCV_QUAD(2) break;
        case 3: DCV_QUAD(3) break;
        default: DCV_QUAD(4) break;
    }
#undef DCV_QUAD
    DCV_NOTE_KERNEL("thin_quad_kernel<%d, %d>", OC, k[0]);
    return true;
}

// which: 0 forward, 1 backward-data, 2 backward-weight
static int conv_dispatch(int which, const dcv_conv_geom* g, const float* a_, const dcv_dims5* xd, const float* w,
                         float* out, const dcv_dims5* yd, int act, float slope, int accumulate,
                         void* ws, size_t ws_bytes, void* stream, size_t* need_only,
                         float* stat = nullptr, size_t stat_bytes = 0, int* stat_parts = nullptr, size_t* stat_need = nullptr,
                         const dcv_wpack* pack = nullptr, size_t* pack_need = nullptr, const float* gate = nullptr, float gate_slope = 0.f) {
    // xd = module input dims, yd = module output dims, always.
    if (g && (g->mfma < 0 || g->mfma > 3)) return fail(DCV_EINVAL, "conv: dcv_conv_geom.mfma must be 0 (process default), 1 (fp32), 2 (bf16 products) or 3 (fp32 on the bf16 pipe)");
    PrecisionScope prec_scope(g);
    // a caller-owned packed copy is valid for ONE effective precision (its format depends on it): never read or fill one stamped for another
    if (pack && pack->buf && !need_only && !pack_need && !stat_need && pack->precision != eff_precision() + 1)
        return fail(DCV_EINVAL, "conv: dcv_wpack.precision is %d but this call runs at precision %d (1 fp32, 2 bf16 products, 3 fp32-on-bf16): %s",
                    pack->precision, eff_precision() + 1, pack->ready ? "a ready pack of another format would be read" : "stamp the buffer with dcv_conv_effective_precision(g)");
    {
        dcv_conv_geom g2;
        dcv_dims5 y2;
        if (!gate && latent_form(g, xd, yd, &g2, &y2)) {
            if (stat_parts) *stat_parts = 0;       // per-(channel, tap) columns are not BatchNorm channels: the BN op runs its own statistics pass
            if (stat_need) { *stat_need = 0; return DCV_OK; }
            return conv_dispatch(which, &g2, a_, xd, w, out, &y2, act, slope, accumulate, ws, ws_bytes, stream, need_only, nullptr, 0, nullptr, nullptr, pack, pack_need);
        }
    }
    int rc = check_geom(g, xd, yd, "conv");
    if (rc != DCV_OK) return rc;
    const int k[3] = {g->kd, g->kh, g->kw}, s[3] = {g->sd, g->sh, g->sw}, p[3] = {g->pd, g->ph, g->pw};
    const int xi[3] = {xd->d, xd->h, xd->w}, yo[3] = {yd->d, yd->h, yd->w};
    const int T = k[0] * k[1] * k[2];
    hipStream_t st = static_cast<hipStream_t>(stream);
    // forward of conv / backward-data of convT : direct gather from the module INPUT side tensor
    const bool direct = (which == 0 && !g->transposed) || (which == 1 && g->transposed);
    if (which == 0 || which == 1) {
        // src tensor dims / dst tensor dims
        const dcv_dims5& src = (which == 0) ? *xd : *yd;
        const dcv_dims5& dst = (which == 0) ? *yd : *xd;
        const int RC = src.c, OC = dst.c;
        std::vector<GatherClass> cls;
        int64_t ws_o, ws_r;
        if (direct) {
            // conv fprop: src = x (gathered at o*s-p+k), dst = y.  weight (cout, cin, T): oc = cout
            // convT dgrad: src = dy (module output), dst = dx; dx[i] = sum dy[i*s-p+k] w[ci,co,k]: weight (cin, cout, T): oc = cin
            const int* o_ext = (which == 0) ? yo : xi;
            const int* in_ext = (which == 0) ? xi : yo;
            cls = direct_classes(k, s, p, o_ext, in_ext);
            ws_o = (int64_t)RC * T;
            ws_r = T;
        } else {
            // conv dgrad: src = dy, dst = dx, weight (cout, cin, T): oc = cin -> ws_o = T, ws_r = cin*T
            // convT fprop: src = x, dst = y, weight (cin, cout, T): oc = cout -> ws_o = T, ws_r = cout*T
            const int* out_ext = (which == 0) ? yo : xi;
            const int* in_ext = (which == 0) ? xi : yo;
            cls = scatter_classes(k, s, p, out_ext, in_ext);
            ws_o = T;
            ws_r = (int64_t)OC * T;
        }
        if (stat_need) {   // upper bound of the fused-BN partial-sum buffer (run_gather may halve the position tile)
            TileCfg tc = pick_gather_tile(OC);
            if (tc.bn >= 64) tc.bm /= 2;
            const int OCp = (OC + tc.bn - 1) / tc.bn * tc.bn;
            int64_t ntm = 0, ncls = 0;
            for (const GatherClass& c : cls) {
                const int64_t Mc = (int64_t)dst.n * std::max(c.o_ext[0], 0) * std::max(c.o_ext[1], 0) * std::max(c.o_ext[2], 0);
                ntm = std::max<int64_t>(ntm, (Mc + tc.bm - 1) / tc.bm);
                ++ncls;
            }
            *stat_need = (size_t)(ncls * ntm * OCp * 2) * sizeof(float);
            return DCV_OK;
        }
        if (pack_need) {
            *pack_need = gather_pack_bytes(RC, OC, cls);
            return DCV_OK;
        }
        const bool head = !stat && !gate && head_shape_ok(k, s, p, g->transposed != 0, *xd, *yd);
        if (need_only) {
            *need_only = gather_ws_bytes(RC, OC, dst.n, cls);
            if (head && which == 0) *need_only = std::max(*need_only, head_fwd_bytes(*xd, k[0]));
            return DCV_OK;
        }
        if (!a_ || !w || !out) return fail(DCV_EINVAL, "conv: null pointer");
        // a discriminator's head, C -> 1 on 8 x 8 planes.  Measured at B = 70 (profiles/r06_ab_heads.txt): the forward kernel wins for the 3-D heads (0.081 -> 0.028 ms,
        // 0.035 -> 0.019) and ties for the 2-D one; the data-gradient kernel wins for the 2-D head (0.046 -> 0.014) and loses to the tile kernel for the 3-D ones
        if (head && which == 0 && k[0] == 4) {
            if (stat_parts) *stat_parts = 0;
            return run_head_fwd(a_, *xd, w, out, *yd, k[0], act, slope, ws, ws_bytes, st);
        }
        if (head && which == 1 && k[0] == 1) {
            if (stat_parts) *stat_parts = 0;
            return run_head_dgrad(a_, *yd, w, out, *xd, k[0], accumulate, st);
        }
        if (!direct && !stat && !gate && try_thin_quad(a_, src, out, dst, w, k, s, p, act, slope, accumulate, st)) {
            if (stat_parts) *stat_parts = 0;
            DCV_LAUNCH_CHECK();
            return DCV_OK;
        }
        return run_gather(a_, src, out, dst, w, RC, OC, ws_o, ws_r, k[1], k[2], cls, act, slope, accumulate, ws, ws_bytes, st,
                          which == 0 ? (g->transposed ? "convT_fwd" : "conv_fwd") : (g->transposed ? "convT_bwd_data" : "conv_bwd_data"),
                          stat, stat_bytes, stat_parts, pack, gate, gate_slope);
    }
    return fail(DCV_EINVAL, "conv: bad dispatch");
}

size_t dcv_conv_workspace_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which) {
    size_t need = 0;
    if (which == 2) {
        dcv_conv_geom g2;
        dcv_dims5 y2;
        if (latent_form(g, x, y, &g2, &y2)) return dcv_conv_workspace_bytes(&g2, x, &y2, 2);
        if (check_geom(g, x, y, "conv_ws") != DCV_OK) return 0;
        const int k[3] = {g->kd, g->kh, g->kw}, s[3] = {g->sd, g->sh, g->sw}, p[3] = {g->pd, g->ph, g->pw};
        const dcv_dims5& D = g->transposed ? *x : *y;
        const dcv_dims5& G = g->transposed ? *y : *x;
        if (run_wgrad(nullptr, D, nullptr, G, nullptr, k, s, p, nullptr, 0, nullptr, "ws", &need) != DCV_OK) return 0;
        return need;
    }
    if (conv_dispatch(which, g, nullptr, x, nullptr, nullptr, y, 0, 0.f, 0, nullptr, 0, nullptr, &need) != DCV_OK) return 0;
    return need;
}

size_t dcv_conv_backward_data_bn_workspace_bytes(const dcv_dims5* dx, int cbn) {
    if (!dx || cbn < 1) return 0;
    // one partial row per workgroup (at most one per 4 output rows of a plane) + the finalize's constants
    return align_up((size_t)dx->n * dx->d * ((dx->h + 3) / 4) * cbn * 2 * sizeof(float), 256) + align_up((size_t)cbn * 2 * sizeof(float), 256) + 256;
}

int dcv_conv_backward_data_bn(const dcv_conv_geom* g, const float* dy, const dcv_dims5* dyd, const float* w, float* dx, const dcv_dims5* dxd, const dcv_wpack* pack,
                              void* ws, size_t ws_bytes, int cbn, const float* bn_x, const dcv_dims5* bn_xd, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, int act, float slope, float* bn_dx, const dcv_dims5* bn_dxd, float* dgamma, float* dbeta,
                              void* ws2, size_t ws2_bytes, int* fused, void* stream) {
    if (!fused || !bn_x || !bn_xd || !gamma || !beta || !save_mean || !save_invstd || !bn_dx || !bn_dxd || !dgamma || !dbeta || !dxd || !ws2)
        return fail(DCV_EINVAL, "conv_backward_data_bn: null pointer");
    *fused = 0;
    if (act != DCV_ACT_NONE && act != DCV_ACT_LEAKY) return fail(DCV_EUNSUPPORTED, "conv_backward_data_bn: the BatchNorm's activation must be none or (Leaky)ReLU");
    if (eff_precision() != 0 && !(g && g->mfma == 1)) return fail(DCV_EUNSUPPORTED, "conv_backward_data_bn: fp32 path only");
    const size_t pbytes = align_up((size_t)dxd->n * dxd->d * ((dxd->h + 3) / 4) * cbn * 2 * sizeof(float), 256);
    if (ws2_bytes < dcv_conv_backward_data_bn_workspace_bytes(dxd, cbn)) return fail(DCV_EWORKSPACE, "conv_backward_data_bn: second workspace too small");
    HeadBnRequest hb;
    memset(&hb, 0, sizeof(hb));
    hb.mode = 1;
    hb.bx = bn_x; hb.bxd = *bn_xd; hb.bdx = bn_dx; hb.bdxd = *bn_dxd;
    hb.gamma = gamma; hb.beta = beta; hb.mean = save_mean; hb.invstd = save_invstd;
    hb.partial = static_cast<float*>(ws2); hb.partial_bytes = pbytes;
    float* cst = reinterpret_cast<float*>(static_cast<char*>(ws2) + pbytes);
    hb.cst = cst;
    hb.cbn = cbn; hb.slope = act == DCV_ACT_LEAKY ? slope : 1.f;
    t_headbn = &hb;
    int rc = conv_dispatch(1, g, dy, dxd, w, dx, dyd, DCV_ACT_NONE, 0.f, 0, ws, ws_bytes, stream, nullptr, nullptr, 0, nullptr, nullptr, pack);
    t_headbn = nullptr;
    if (rc != DCV_OK || !hb.used) return rc;      // not the head's geometry: the plain data gradient ran, dx is complete, *fused stays 0
    const double count = (double)bn_xd->n * bn_xd->d * bn_xd->h * bn_xd->w;
    hipLaunchKernelGGL(head_bn_finalize_kernel, dim3(cbn), dim3(256), 0, static_cast<hipStream_t>(stream), hb.partial, hb.nwg, cbn, count, gamma, save_invstd, dgamma, dbeta, cst);
    DCV_LAUNCH_CHECK();
    hb.mode = 2;
    hb.used = 0;
    dcv_wpack pk2;
    const dcv_wpack* pk2p = nullptr;
    if (pack && pack->buf) { pk2 = *pack; pk2.ready = 1; pk2p = &pk2; }      // the first call packed the weights
    t_headbn = &hb;
    rc = conv_dispatch(1, g, dy, dxd, w, dx, dyd, DCV_ACT_NONE, 0.f, 0, ws, ws_bytes, stream, nullptr, nullptr, 0, nullptr, nullptr, pk2p);
    t_headbn = nullptr;
    if (rc != DCV_OK) return rc;
    if (!hb.used) return fail(DCV_EINVAL, "conv_backward_data_bn: internal: the second pass did not take the fused kernel");
    *fused = 1;
    return DCV_OK;
}

size_t dcv_conv_packed_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which) {
    size_t need = 0;
    if (which != 0 && which != 1) return 0;
    if (conv_dispatch(which, g, nullptr, x, nullptr, nullptr, y, 0, 0.f, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, &need) != DCV_OK) return 0;
    return need;
}

int dcv_conv_forward(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd, const float* w, float* y, const dcv_dims5* yd,
                     int act, float slope, const dcv_wpack* pack, void* ws, size_t ws_bytes, void* stream) {
    return conv_dispatch(0, g, x, xd, w, y, yd, act, slope, 0, ws, ws_bytes, stream, nullptr, nullptr, 0, nullptr, nullptr, pack);
}

size_t dcv_conv_stats_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y) {
    size_t need = 0;
    if (conv_dispatch(0, g, nullptr, x, nullptr, nullptr, y, 0, 0.f, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, &need) != DCV_OK) return 0;
    return need;
}

int dcv_conv_forward_stats(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd, const float* w, float* y, const dcv_dims5* yd,
                           float* stat, size_t stat_bytes, int* nparts, int* pitch, const dcv_wpack* pack, void* ws, size_t ws_bytes, void* stream) {
    if (!stat || !nparts || !pitch) return fail(DCV_EINVAL, "conv_forward_stats: null pointer");
    const TileCfg tc = pick_gather_tile(yd ? yd->c : 1);
    *pitch = yd ? (yd->c + tc.bn - 1) / tc.bn * tc.bn : 0;
    return conv_dispatch(0, g, x, xd, w, y, yd, DCV_ACT_NONE, 0.f, 0, ws, ws_bytes, stream, nullptr, stat, stat_bytes, nparts, nullptr, pack);
}

int dcv_conv_backward_data(const dcv_conv_geom* g, const float* dy, const dcv_dims5* dyd, const float* w, float* dx, const dcv_dims5* dxd,
                           int accumulate, const dcv_wpack* pack, void* ws, size_t ws_bytes, void* stream) {
    return conv_dispatch(1, g, dy, dxd, w, dx, dyd, DCV_ACT_NONE, 0.f, accumulate, ws, ws_bytes, stream, nullptr, nullptr, 0, nullptr, nullptr, pack);
}

int dcv_conv_backward_data_gated(const dcv_conv_geom* g, const float* dy, const dcv_dims5* dyd, const float* w, float* dx, const dcv_dims5* dxd,
                                 int accumulate, const float* x, const dcv_dims5* xd, int act, float slope, const dcv_wpack* pack,
                                 void* ws, size_t ws_bytes, void* stream) {
    if (!x || !xd || !dxd) return fail(DCV_EINVAL, "conv_backward_data_gated: null pointer");
    if (act != DCV_ACT_LEAKY) return fail(DCV_EUNSUPPORTED, "conv_backward_data_gated: only (Leaky)ReLU derivatives can be read off the input");
    if (!same_shape(*xd, *dxd) || xd->sn != dxd->sn || xd->sc != dxd->sc || xd->sd != dxd->sd || xd->sh != dxd->sh || xd->sw != dxd->sw)
        return fail(DCV_EUNSUPPORTED, "conv_backward_data_gated: x and dx must share shape and strides");
    return conv_dispatch(1, g, dy, dxd, w, dx, dyd, DCV_ACT_NONE, 0.f, accumulate, ws, ws_bytes, stream, nullptr, nullptr, 0, nullptr, nullptr, pack, nullptr, x, slope);
}

int dcv_conv_backward_weight(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd, const float* dy, const dcv_dims5* dyd,
                             float* dw, void* ws, size_t ws_bytes, void* stream) {
    if (g && (g->mfma < 0 || g->mfma > 3)) return fail(DCV_EINVAL, "conv_bwd_weight: dcv_conv_geom.mfma must be 0, 1, 2 or 3");
    PrecisionScope prec_scope(g);
    {
        dcv_conv_geom g2;
        dcv_dims5 y2;
        if (latent_form(g, xd, dyd, &g2, &y2)) return dcv_conv_backward_weight(&g2, x, xd, dy, &y2, dw, ws, ws_bytes, stream);
    }
    int rc = check_geom(g, xd, dyd, "conv_bwd_weight");
    if (rc != DCV_OK) return rc;
    if (!x || !dy || !dw || !ws) return fail(DCV_EINVAL, "conv_bwd_weight: null pointer");
    const int k[3] = {g->kd, g->kh, g->kw}, s[3] = {g->sd, g->sh, g->sw}, p[3] = {g->pd, g->ph, g->pw};
    // conv : dw[co][ci][k] = sum dy[n,co,o] x[n,ci,o*s-p+k]  -> dense = dy, gathered = x
    // convT: dw[ci][co][k] = sum x[n,ci,i] dy[n,co,i*s-p+k]  -> dense = x,  gathered = dy
    if (!g->transposed)
        return run_wgrad(dy, *dyd, x, *xd, dw, k, s, p, ws, ws_bytes, static_cast<hipStream_t>(stream), "conv_bwd_weight", nullptr);
    return run_wgrad(x, *xd, dy, *dyd, dw, k, s, p, ws, ws_bytes, static_cast<hipStream_t>(stream), "convT_bwd_weight", nullptr);
}

// The geometry BOTH normalise-on-load kernels take (the forward must only leave the BatchNorm output unwritten where the weight gradient can do without it too):
// the 3x3 / 1 / 1 transposed head with 3 output channels on 64-wide rows, 128 k input channels, whole 32-channel BatchNorm groups, 16-byte aligned BatchNorm input
static bool head_on_load_ok(const dcv_conv_geom* g, const dcv_dims5* xd, const dcv_dims5* yd, int cbn, const float* bn_x, const dcv_dims5* bn_xd) {
    return g->transposed && g->kd == 1 && g->kh == 3 && g->kw == 3 && g->sd == 1 && g->sh == 1 && g->sw == 1 && g->pd == 0 && g->ph == 1 && g->pw == 1 &&
           yd->c == 3 && xd->c % 128 == 0 && xd->w == 64 && yd->w == 64 && xd->h == yd->h && xd->h % 4 == 0 && xd->sw == 1 && (g->mfma == 0 || g->mfma == 1) &&
           cbn % 32 == 0 && cbn > 0 && cbn <= xd->c && bn_xd && bn_xd->sc % 4 == 0 && bn_xd->sh % 4 == 0 && bn_xd->sn % 4 == 0 && bn_xd->sh == xd->sh &&
           (reinterpret_cast<uintptr_t>(bn_x) & 15) == 0 && xd->sc % 4 == 0 && xd->sh % 4 == 0 && xd->sn % 4 == 0;
}

static int bn_view_of(BnView* v, int cbn, const float* bn_x, const dcv_dims5* bn_xd, const float* gamma, const float* beta, const float* mean, const float* invstd,
                      int bn_act, float bn_slope, const dcv_dims5* xd, const char* tag) {
    if (!bn_x || !bn_xd || !gamma || !beta || !mean || !invstd || !xd) return fail(DCV_EINVAL, "%s: null pointer", tag);
    if (bn_act != DCV_ACT_NONE && bn_act != DCV_ACT_LEAKY) return fail(DCV_EUNSUPPORTED, "%s: the BatchNorm's activation must be none or (Leaky)ReLU", tag);
    if (eff_precision() != 0) return fail(DCV_EUNSUPPORTED, "%s: fp32 path only", tag);
    if (cbn < 1 || cbn > xd->c || bn_xd->c != cbn || bn_xd->n != xd->n || bn_xd->d != 1 || xd->d != 1 || bn_xd->h != xd->h || bn_xd->w != xd->w || bn_xd->sw != 1 ||
        bn_xd->sn < 0 || bn_xd->sc < 0 || bn_xd->sh < 0 || (int64_t)cbn * bn_xd->sc + (int64_t)bn_xd->h * bn_xd->sh >= (1ll << 29))
        return fail(DCV_EUNSUPPORTED, "%s: the BatchNorm input must match the operand's first channels (2-D, unit column stride)", tag);
    memset(v, 0, sizeof(*v));
    v->bx = bn_x; v->gamma = gamma; v->beta = beta; v->mean = mean; v->invstd = invstd;
    v->bx_sn = bn_xd->sn; v->bx_sc = (int32_t)bn_xd->sc; v->bx_sh = (int32_t)bn_xd->sh; v->cbn = cbn; v->act = bn_act; v->slope = bn_slope;
    return DCV_OK;
}

int dcv_conv_forward_bn(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd, const float* w, float* y, const dcv_dims5* yd, int act, float slope,
                        const dcv_wpack* pack, void* ws, size_t ws_bytes, int cbn, const float* bn_x, const dcv_dims5* bn_xd, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, int bn_act, float bn_slope, void* stream) {
    // only the RGB head's geometry has the kernel: refuse everything else BEFORE anything runs (the caller then materialises the BatchNorm output: dcv_bn_apply)
    if (!g || !xd || !yd) return fail(DCV_EINVAL, "conv_forward_bn: null pointer");
    if (!head_on_load_ok(g, xd, yd, cbn, bn_x, bn_xd) || (x && (reinterpret_cast<uintptr_t>(x) & 15) != 0))
        return fail(DCV_EUNSUPPORTED, "conv_forward_bn: not the 3-channel 3x3 transposed head on 64-wide rows with 128 k input channels");
    BnView v;
    int rc = bn_view_of(&v, cbn, bn_x, bn_xd, gamma, beta, save_mean, save_invstd, bn_act, bn_slope, xd, "conv_forward_bn");
    if (rc != DCV_OK) return rc;
    t_bnview = &v;
    rc = conv_dispatch(0, g, x, xd, w, y, yd, act, slope, 0, ws, ws_bytes, stream, nullptr, nullptr, 0, nullptr, nullptr, pack);
    t_bnview = nullptr;
    if (rc == DCV_OK && !v.used) return fail(DCV_EINVAL, "conv_forward_bn: internal: the dispatch did not reach the head's kernel (the output was computed from an unwritten operand)");
    return rc;
}

int dcv_conv_backward_weight_bn(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd, const float* dy, const dcv_dims5* dyd, float* dw, int accumulate,
                                void* ws, size_t ws_bytes, int cbn, const float* bn_x, const dcv_dims5* bn_xd, const float* gamma, const float* beta,
                                const float* save_mean, const float* save_invstd, int bn_act, float bn_slope, void* stream) {
    if (!g || !xd || !dyd) return fail(DCV_EINVAL, "conv_backward_weight_bn: null pointer");
    if (!head_on_load_ok(g, xd, dyd, cbn, bn_x, bn_xd))
        return fail(DCV_EUNSUPPORTED, "conv_backward_weight_bn: not the 3-channel 3x3 transposed head on 64-wide rows with 128 k input channels");
    BnView v;
    int rc = bn_view_of(&v, cbn, bn_x, bn_xd, gamma, beta, save_mean, save_invstd, bn_act, bn_slope, xd, "conv_backward_weight_bn");
    if (rc != DCV_OK) return rc;
    t_bnview = &v;
    t_wgrad_acc = accumulate ? 1 : 0;
    rc = dcv_conv_backward_weight(g, x, xd, dy, dyd, dw, ws, ws_bytes, stream);
    t_wgrad_acc = 0;
    t_bnview = nullptr;
    if (rc == DCV_OK && !v.used) return fail(DCV_EINVAL, "conv_backward_weight_bn: internal: the dispatch did not reach the head's kernel");
    return rc;
}

// dw = (accumulate ? dw : 0) + corr(x, dy): a weight used twice in one backward (a discriminator on the real and the fake batch, trainer.py:299-309) or whose
// .grad already holds an earlier backward's gradient (the G phase on top of the D phase, trainer.py:356 after :319) gets its sum formed by the slab reduce
// instead of by a separate elementwise add
int dcv_conv_backward_weight_acc(const dcv_conv_geom* g, const float* x, const dcv_dims5* xd, const float* dy, const dcv_dims5* dyd, float* dw, int accumulate,
                                 void* ws, size_t ws_bytes, void* stream) {
    t_wgrad_acc = accumulate ? 1 : 0;
    const int rc = dcv_conv_backward_weight(g, x, xd, dy, dyd, dw, ws, ws_bytes, stream);
    t_wgrad_acc = 0;
    return rc;
}

}  // extern "C"
