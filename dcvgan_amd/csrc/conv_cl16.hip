// bf16 channels-last ("CL16") implicit-GEMM convolutions for gfx950 on v_mfma_f32_32x32x16_bf16 — the bf16 DATA path of
// BASELINE configs[2] ("surreal-depth1, bf16 MFMA") and configs[4] ("fp16 MFMA"; bf16 has the same MFMA rate and fragment path and
// fp32's exponent range).  The fp32 NCDHW kernels of conv_mfma.hip stay the default and the parity path; here
//
//   * activations and their gradients live in HBM as bf16 with the CHANNEL innermost (memory order n, d, h, w, c; the channel
//     stride is 1 and the pixel pitch `ldc` a multiple of 8, so a concatenation is two channel ranges of one buffer),
//   * weights are fp32 masters in torch layout, packed per optimiser step into bf16 K-major tiles,
//   * accumulation, BatchNorm statistics, weight gradients and the optimiser are fp32.
//
// With K = (tap, channel) contiguous per pixel an MFMA operand fragment (8 consecutive k of one row) is ONE 16-byte LDS read and
// one 16-byte LDS-DMA granule per lane, for every stride / padding / tap count: the gathering is done by per-lane global
// addresses, zero padding by the buffer range check.
//
//  gather GEMM   Y[m, oc] = sum_{t, c} X[pos(m, t), c] * Wp[t, c, oc]     forward, data gradient (one launch, z = stride-parity class)
//      MFMA A = weights (rows = oc), B = activations (cols = positions): an accumulator register quad holds 4 consecutive oc of one
//      position = one 8-byte channels-last store.
//  wgrad GEMM    R[t, dc, gc] = sum_m D[m, dc] * G[pos(m, t), gc]          K = positions: both operands are read from their
//      [position][channel] LDS images with the transposing read ds_read_b64_tr_b16.
#include "dcv_common.h"

#ifdef DCV_CL_FP16      // the fp16 build of this file: same code, element type cl_h = _Float16, entry points dcv_clf16_*
#define dcv_cl_debug_read_stamps dcv_clf16_debug_read_stamps
#define dcv_cl_packed_bytes dcv_clf16_packed_bytes
#define dcv_cl_conv_workspace_bytes dcv_clf16_conv_workspace_bytes
#define dcv_cl_pack_weights dcv_clf16_pack_weights
#define dcv_cl_conv_forward dcv_clf16_conv_forward
#define dcv_cl_conv_stats_bytes dcv_clf16_conv_stats_bytes
#define dcv_cl_conv_forward_stats dcv_clf16_conv_forward_stats
#define dcv_cl_conv_backward_data dcv_clf16_conv_backward_data
#define dcv_cl_conv_backward_data_gated dcv_clf16_conv_backward_data_gated
#define dcv_cl_wgrad_workspace_bytes dcv_clf16_wgrad_workspace_bytes
#define dcv_cl_conv_backward_weight dcv_clf16_conv_backward_weight
#define dcv_cl_conv_backward_weight_acc dcv_clf16_conv_backward_weight_acc
#endif

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace dcv {
extern thread_local char g_last_kernel[160];
#ifdef DCV_CL_FP16
inline namespace clf16 {
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;


// --------------------------------------------------------------------------- //
// geometry: one "class" = the output positions that share a tap set (all of them for a direct gather; one stride-parity class
// of a scatter-form op), as in conv_mfma.hip
// --------------------------------------------------------------------------- //
struct ClDim {
    int32_t n;                 // taps along this dim (<= 4)
    int32_t mul, base, size;   // gathered coordinate = o * mul + base + delta[u], valid in [0, size)
    int32_t delta[4];
    int32_t kidx[4];           // filter index of tap u
};
struct ClClass {
    int32_t o_ext[3], out_mul[3], out_off[3];
    ClDim t[3];
};

static std::vector<ClClass> cl_direct_classes(const int k[3], const int s[3], const int p[3], const int o_ext[3], const int in_ext[3]) {
    ClClass c;
    memset(&c, 0, sizeof(c));
    for (int d = 0; d < 3; ++d) {
        c.o_ext[d] = o_ext[d]; c.out_mul[d] = 1; c.out_off[d] = 0;
        c.t[d].n = k[d]; c.t[d].mul = s[d]; c.t[d].base = -p[d]; c.t[d].size = in_ext[d];
        for (int u = 0; u < k[d]; ++u) { c.t[d].delta[u] = u; c.t[d].kidx[u] = u; }
    }
    return {c};
}
// scatter form (conv data gradient / transposed conv forward): out[o] += src[i] w[k] with o = i s - p + k  =>  per class cls = o mod s:
// k = k0 + u s, i = (o + p - k) / s = o' + q - u  (o = o' s + cls)
static std::vector<ClClass> cl_scatter_classes(const int k[3], const int s[3], const int p[3], const int out_ext[3], const int in_ext[3]) {
    std::vector<ClClass> out;
    for (int cd = 0; cd < s[0]; ++cd)
        for (int ch = 0; ch < s[1]; ++ch)
            for (int cw = 0; cw < s[2]; ++cw) {
                const int cls[3] = {cd, ch, cw};
                ClClass c;
                memset(&c, 0, sizeof(c));
                for (int d = 0; d < 3; ++d) {
                    c.o_ext[d] = (out_ext[d] - cls[d] + s[d] - 1) / s[d];
                    c.out_mul[d] = s[d];
                    c.out_off[d] = cls[d];
                    const int k0 = (cls[d] + p[d]) % s[d];
                    const int q = (cls[d] + p[d] - k0) / s[d];
                    c.t[d].mul = 1; c.t[d].base = q; c.t[d].size = in_ext[d];
                    int u = 0;
                    for (int kk = k0; kk < k[d] && u < 4; kk += s[d], ++u) { c.t[d].delta[u] = -u; c.t[d].kidx[u] = kk; }
                    c.t[d].n = u;
                }
                out.push_back(c);
            }
    return out;
}

static inline int pad8(int c) { return (c + 7) / 8 * 8; }
// K granularity of a gathered operand with C channels: thin (C <= 8: one 16-byte granule per tap, 4 taps per 32-deep K step) or
// whole 32-channel blocks
static inline bool cl_thin(int C) { return C <= 8; }
static inline int cl_cp(int C) { return cl_thin(C) ? 8 : (C + 31) / 32 * 32; }

// --------------------------------------------------------------------------- //
// gather GEMM
// --------------------------------------------------------------------------- //
struct ClGatherArgs {
    const cl_h* x;
    cl_h* y;
    const cl_h* wp;            // [step][OCp][32] bf16
    int32_t M, OCp, nsteps, T;   // positions, padded output channels (multiple of the tile's), 32-deep K steps, taps
    int32_t cblk, y_c;           // 32-channel blocks per tap (0 = thin); channels to store (destination channels, multiple of 4)
    int32_t x_cmax, coalesce;    // bytes of one pixel's own channels, rounded up to a granule (granules past them read as zeros); epilogue through LDS (16-byte row-order stores)
    FastDiv div_sp, div_hw, div_w;
    ClDim td, th, tw;
    int64_t x_sn, y_sn;          // element strides
    int32_t x_sd, x_sh, x_sw;
    int32_t y_sd, y_sh, y_sw, y_off;   // output strides incl. the class multiplier, class offset
    int32_t act, accumulate;
    float slope, pad1;
    uint32_t x_bytes, y_bytes;   // buffer extents (range check = padding and guard)
    int32_t toff[64];            // byte offset of tap t relative to the position's base pixel
    int32_t tsel[64];            // ud | uh << 2 | uw << 4
    // BatchNorm sums of the stored outputs, left by the epilogue (conv -> BatchNorm pairs, forward, no activation): stat[stat_row0 + position tile][OCp][2] = {sum, sum of squares}
    float* stat;
    int32_t stat_row0, pad2;
    // gated data gradient (dcv_cl_conv_backward_data_gated): the (Leaky)ReLU derivative of the layer that PRODUCED this convolution's input, read off that input —
    // a tensor of the destination's shape and strides — and applied in the epilogue: dx = (accumulate ? dx : 0) + conv^T(dy, w), then dx *= (gate > 0 ? 1 : gate_slope)
    const cl_h* gate;
    float gate_slope; int32_t pad3;
    // split-K (few position tiles, long K loop: the latent layers at 1x1 ... 4x4 positions per image): blockIdx.y handles K steps [y ks_per, (y + 1) ks_per) and leaves
    // raw fp32 partial sums in slab[y][position][OCp]; cl_splitk_reduce_kernel adds the slabs in order (bitwise reproducible), applies the activation and stores bf16
    float* slab;
    int32_t ks_per, slab_m;      // K steps per split; positions per slab (position tiles x BM)
};
struct ClGatherPack {
    ClGatherArgs c[4];
    int32_t ncls, tiles_oc, tiles_m, pad;
};

__device__ __forceinline__ float cl_act(float v, int act, float slope) {
    if (act == DCV_ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (act == DCV_ACT_TANH) return tanhf(v);
    return v;
}

// sum over each 32-lane half of the wave, valid in lanes 16-31 / 48-63 (five DPP adds; the fp32 path's half_wave_sum)
__device__ __forceinline__ float cl_half_wave_sum(float v) {
#define CL_DPP(X, CTRL, ROWS) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, X), CTRL, ROWS, 0xf, false))
    v += CL_DPP(v, 0xB1, 0xf);     // quad_perm [1,0,3,2]
    v += CL_DPP(v, 0x4E, 0xf);     // quad_perm [2,3,0,1]
    v += CL_DPP(v, 0x141, 0xf);    // row_half_mirror
    v += CL_DPP(v, 0x140, 0xf);    // row_mirror
    v += CL_DPP(v, 0x142, 0xa);    // row_bcast:15 into rows 1 and 3
#undef CL_DPP
    return v;
}

__device__ __forceinline__ uint32_t cl_dim_mask(const ClDim& t, int o) {
    uint32_t m = 0;
    const int p0 = o * t.mul + t.base;
#pragma unroll
    for (int u = 0; u < 4; ++u) m |= ((unsigned)(p0 + t.delta[u]) < (unsigned)t.size && u < t.n) ? (1u << u) : 0u;
    return m;
}

// 256 threads = 4 waves laid out WOC x WM, each wave TOC x TM MFMA tiles of 32 x 32; BN = 32 TOC WOC output channels, BM = 32 TM WM
// positions, K step 32.  LDS: two stages of [BM rows][64 B] activations + [BN rows][64 B] weights; the four 16-byte chunks of a
// row are XOR-swizzled with (row >> 2) & 3 on the DMA's SOURCE side (an LDS-DMA lands lane-linear), which makes the fragment
// reads "32 rows x one chunk" (ds_read_b128) conflict-free.
#ifdef DCV_CL_STAMP
// cycle-stamped build (tools/build_stamp_cl.sh, tools/stamps_cl.py): per wave of the first 4096 workgroups {prologue, wait + barrier, DMA issue, fragment reads + MFMA issue,
// epilogue, lifetime, steps} in shader-clock cycles
__device__ unsigned long long g_cl_stamps[4096][8][8];
#define CL_T() __builtin_readcyclecounter()
#else
#define CL_T() 0ull
#endif
template <int N>
__device__ __forceinline__ void cl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NS = LDS stages: the DMAs of K step st + NS - 1 are issued at the top of step st (NS = 2: one step ahead, four workgroups per CU; NS = 3: two steps
// ahead behind a counted vmcnt wait, three workgroups per CU)
template <int TOC, int TM, int WOC, int WM, bool THIN, int NS>
__global__ __launch_bounds__(64 * WOC * WM, WOC * WM == 4 ? 2 : 1) void cl_gather_kernel(const ClGatherPack pack) {
    constexpr int BN = 32 * TOC * WOC, BM = 32 * TM * WM, NT = 64 * WOC * WM;      // NT threads: 4 waves, or 8 (128 x 256 tile: 3 LDS-DMA granules per wave and 8 MFMAs instead of 4)
    constexpr int XPT = BM * 4 / NT, WPT = (BN * 4 + NT - 1) / NT;
    static_assert((BM * 4) % NT == 0 && (BM * (BN / 8)) % NT == 0 && BM <= NT, "tile / thread-count combination");
    constexpr int XB = BM * 64, WB = BN * 64, STAGE = XB + WB;
    static_assert(WOC * WM == 4 || WOC * WM == 8, "4 or 8 waves");
    static_assert(NS >= 2 && NS <= 4, "2-4 stages");
    // epilogue image: the tile as [position][BN channels] bf16, pitch BN * 2 + 16 bytes, + one byte offset per position row
    constexpr int EPITCH = BN * 2 + 16, EPI = BM * EPITCH + BM * 4;
    constexpr int SMEM = (NS * STAGE + (THIN ? 256 : 0)) > EPI ? (NS * STAGE + (THIN ? 256 : 0)) : EPI;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    // workgroup -> (class, oc tile, m tile): ids 8 apart (the same XCD's L2) share the gathered operand
    const unsigned grp = (unsigned)(pack.tiles_oc * pack.ncls), loc = blockIdx.x >> 3;
    const unsigned gi = loc % grp;
    const int m_t = (int)((loc / grp) * 8 + (blockIdx.x & 7));
    const int oc_t = (int)(gi % (unsigned)pack.tiles_oc);
    const ClGatherArgs& a = pack.c[gi / (unsigned)pack.tiles_oc];
    const int m0 = m_t * BM, oc0 = oc_t * BN;
    if (m0 >= a.M) return;
    [[maybe_unused]] const unsigned long long ts_start = CL_T();
    [[maybe_unused]] unsigned long long ts_wait = 0, ts_issue = 0, ts_mma = 0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int woc = wave / WM, wm = wave % WM;
    const int l31 = lane & 31, lhi = lane >> 5;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cl_h*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cl_h*>(a.wp), 0, 0x80000000u, 0x00020000);
    // thin operands: a lane's granule is one TAP of the step, so the tap offset is per lane: table in LDS (one array: see the LDS-DMA tracking
    // note in conv_mfma.hip), read one step ahead of its use
    const int32_t* toff_l = reinterpret_cast<const int32_t*>(smem + NS * STAGE);
    if constexpr (THIN) {
        if (tid < 64) reinterpret_cast<int32_t*>(smem + NS * STAGE)[tid] = a.toff[tid];
        __syncthreads();
    }

    // ---- staging roles: granule g = tid + 256 s -> row g >> 2, physical chunk g & 3, logical chunk (g & 3) ^ ((row >> 2) & 3)
    uint32_t xbase[XPT];
    uint64_t xmask[XPT];
    int xchunk[XPT];
#pragma unroll
    for (int s = 0; s < XPT; ++s) {
        const int g = tid + NT * s, row = g >> 2, c = (g & 3) ^ ((row >> 2) & 3);
        const int m = m0 + row;
        xchunk[s] = c;
        uint64_t vm = 0;
        uint32_t base = 0;
        if (m < a.M) {
            const uint32_t n = fdiv((uint32_t)m, a.div_sp);
            uint32_t r = (uint32_t)m - n * a.div_sp.div;
            const uint32_t od = fdiv(r, a.div_hw);
            r -= od * a.div_hw.div;
            const uint32_t oh = fdiv(r, a.div_w);
            const uint32_t ow = r - oh * a.div_w.div;
            const uint32_t md = cl_dim_mask(a.td, (int)od), mh = cl_dim_mask(a.th, (int)oh), mw = cl_dim_mask(a.tw, (int)ow);
            // signed pixel offset of tap (0,0,0)'s base; every VALID tap's offset makes the sum non-negative
            const int64_t e = (int64_t)n * a.x_sn + (int64_t)((int)od * a.td.mul + a.td.base) * a.x_sd + (int64_t)((int)oh * a.th.mul + a.th.base) * a.x_sh +
                              (int64_t)((int)ow * a.tw.mul + a.tw.base) * a.x_sw;
            base = (uint32_t)(2 * e) + (THIN ? 0u : (uint32_t)(16 * c));
            // bit t = tap t in range, t = (ud, uh, uw) row-major as the host enumerates them (tsel).  Nested loops over the per-dim counts: no table load per
            // tap (the tsel[t] scalar loads of the first form were a chain of ~250-cycle latencies: 16 k cycles of prologue at 64 taps; cycle stamps, round 5)
            {
                int t = 0;
                for (int ud = 0; ud < a.td.n; ++ud) {
                    const uint32_t bd = (md >> ud) & 1u;
                    for (int uh = 0; uh < a.th.n; ++uh) {
                        const uint32_t bh = bd & (mh >> uh);
                        for (int uw = 0; uw < a.tw.n; ++uw, ++t) vm |= (uint64_t)(bh & (mw >> uw) & 1u) << t;
                    }
                }
            }
        }
        xbase[s] = base;
        xmask[s] = vm;
    }
    uint32_t wvo[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int g = tid + NT * j, row = g >> 2, c = (g & 3) ^ ((row >> 2) & 3);
        wvo[j] = (uint32_t)(((oc0 + row) * 32 + c * 8) * 2);
    }
    const uint32_t wstep = (uint32_t)a.OCp * 64u;

#if defined(__HIP_DEVICE_COMPILE__)
#define CL_ISSUE(STEP, BUF)                                                                                                   \
    {                                                                                                                         \
        const int st_ = (STEP);                                                                                               \
        char* xb_ = smem + (BUF) * STAGE;                                                                                     \
        char* wb_ = xb_ + XB;                                                                                                 \
        if constexpr (THIN) {                                                                                                 \
            _Pragma("unroll") for (int s = 0; s < XPT; ++s) {                                                                 \
                const int t_ = st_ * 4 + xchunk[s];                                                                           \
                const uint32_t ok_ = (uint32_t)(xmask[s] >> t_) & 1u;                                                         \
                const uint32_t vo_ = ok_ ? xbase[s] + (uint32_t)toff_l[t_ & 63] : 0xffffffffu;                               \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(xb_ + (s * NT + wave * 64) * 16), 16, vo_, 0, 0, 0); \
            }                                                                                                                 \
        } else {                                                                                                              \
            /* the per-lane offsets change with the TAP only: (re)formed at a tap's first channel block; the channel block rides on the scalar offset */ \
            if (nx_cb == 0 || ragged || xvo_stale) {      /* (xvo_stale: a split-K share may start in the middle of a tap) */             \
                xvo_stale = false;                                                                                            \
                const uint32_t to_ = to_cur;                                                                                  \
                _Pragma("unroll") for (int s = 0; s < XPT; ++s) {                                                             \
                    /* granules past the operand's own channels (a 16- or 24-channel slice of a wider buffer) are padding, not the neighbour's data */ \
                    const uint32_t ok_ = ((uint32_t)(xmask[s] >> nx_tap) & 1u) & (uint32_t)(nx_cb * 64 + 16 * xchunk[s] < a.x_cmax); \
                    xvo[s] = ok_ ? xbase[s] + to_ : 0xffffffffu;                                                              \
                }                                                                                                             \
            }                                                                                                                 \
            _Pragma("unroll") for (int s = 0; s < XPT; ++s)                                                                   \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(xb_ + (s * NT + wave * 64) * 16), 16, xvo[s], nx_cb * 64, 0, 0); \
            if (++nx_cb == a.cblk) { nx_cb = 0; ++nx_tap; to_cur = (uint32_t)__builtin_amdgcn_readlane(toff_v, nx_tap & 63); }   /* lane t holds tap t's offset: no memory access in the loop */ \
        }                                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < WPT; ++j)                                                                       \
            if ((BN * 4) % NT == 0 || wave * 64 + NT * j < BN * 4)                                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_t*)(wb_ + (j * NT + wave * 64) * 16), 16, wvo[j], st_ * wstep, 0, 0); \
    }
#else
#define CL_ISSUE(STEP, BUF) { (void)wstep; (void)wvo; (void)xbase; (void)xmask; (void)xchunk; (void)xrs; (void)wrs; (void)toff_l; (void)xvo; (void)nx_tap; (void)nx_cb; (void)ragged; (void)to_cur; (void)toff_v; (void)xvo_stale; }
#endif
    // issue state of the thick form: K steps are issued in order, so (tap, channel block) of the next one are counters, not a division per step
    [[maybe_unused]] uint32_t xvo[XPT];
    [[maybe_unused]] int nx_tap = 0, nx_cb = 0;
    [[maybe_unused]] bool xvo_stale = true;
    // tap offsets: lane t keeps toff[t]; the loop picks the next tap's with v_readlane (a scalar load of toff[tap] at every tap boundary stalled the wave ~250 cycles)
    [[maybe_unused]] const int toff_v = a.toff[lane];
    [[maybe_unused]] uint32_t to_cur = (uint32_t)a.toff[0];
#ifdef DCV_CL_NO_VO_CACHE
    [[maybe_unused]] const bool ragged = true;      // A/B build: offsets re-formed at every K step
#else
    [[maybe_unused]] const bool ragged = a.x_cmax < a.cblk * 64;      // the last channel block is partly padding: granule validity then depends on the block
#endif

    f32x16 acc[TOC][TM];
#pragma unroll
    for (int i = 0; i < TOC; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: row r, K16 half s, lane half lhi -> logical chunk 2 s + lhi, physical chunk ^ ((r >> 2) & 3)
    uint32_t aoff[TOC][2], boff[TM][2];
#pragma unroll
    for (int i = 0; i < TOC; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int row = (woc * TOC + i) * 32 + l31;
            aoff[i][s] = (uint32_t)(XB + row * 64 + (((2 * s + lhi) ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int row = (wm * TM + j) * 32 + l31;
            boff[j][s] = (uint32_t)(row * 64 + (((2 * s + lhi) ^ ((row >> 2) & 3)) << 4));
        }

    // K steps of this workgroup: all of them, or split blockIdx.y's share
    const int st0 = a.slab ? (int)blockIdx.y * a.ks_per : 0;
    const int nst = a.slab ? min(a.nsteps, st0 + a.ks_per) : a.nsteps;
    if (st0 > 0) {      // issue state of the thick form at an arbitrary first step
        if constexpr (!THIN) { nx_tap = st0 / a.cblk; nx_cb = st0 - nx_tap * a.cblk; to_cur = (uint32_t)__builtin_amdgcn_readlane(toff_v, nx_tap & 63); }
    }
    // vector-memory instructions one K step costs THIS wave (the weight tile of a 32-channel tile is 128 granules: waves 0-1 only)
    const bool wfull = (BN * 4) % NT == 0 || wave * 64 + NT * (WPT - 1) < BN * 4;
    CL_ISSUE(st0, 0)
    if constexpr (NS >= 3) { if (nst > st0 + 1) CL_ISSUE(st0 + 1, 1) }
    if constexpr (NS >= 4) { if (nst > st0 + 2) CL_ISSUE(st0 + 2, 2) }
    [[maybe_unused]] const unsigned long long ts_loop = CL_T();
    int buf = 0;
    [[maybe_unused]] int nbuf = NS - 1;      // stage of step st; stage the step issued now lands in
    for (int st = st0; st < nst; ++st) {
        [[maybe_unused]] const unsigned long long t0_ = CL_T();
        // step st's granules have landed: all but the (NS - 2) younger steps' instructions of this wave are done
        if constexpr (NS == 2) cl_wait_vm<0>();
        else {
            const int ahead = min(NS - 2, nst - 1 - st);
            if (ahead == 0) cl_wait_vm<0>();
            else if (ahead == 1) { if (wfull) cl_wait_vm<XPT + WPT>(); else cl_wait_vm<XPT + WPT - 1>(); }
            else { if (wfull) cl_wait_vm<2 * (XPT + WPT)>(); else cl_wait_vm<2 * (XPT + WPT - 1)>(); }
        }
        __syncthreads();
        [[maybe_unused]] const unsigned long long t1_ = CL_T();
        if (st + NS - 1 < nst) CL_ISSUE(st + NS - 1, nbuf)
        [[maybe_unused]] const unsigned long long t2_ = CL_T();
        const char* sb = smem + buf * STAGE;
        nbuf = buf;
        buf = buf + 1 == NS ? 0 : buf + 1;
        cl_h8 a8[TOC][2], b8[TM][2];
#pragma unroll
        for (int i = 0; i < TOC; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) a8[i][s] = *reinterpret_cast<const cl_h8*>(sb + aoff[i][s]);
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) b8[j][s] = *reinterpret_cast<const cl_h8*>(sb + boff[j][s]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < TOC; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = CL_MFMA(a8[i][s], b8[j][s], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
#ifdef DCV_CL_STAMP
        { const unsigned long long t3_ = CL_T(); ts_wait += t1_ - t0_; ts_issue += t2_ - t1_; ts_mma += t3_ - t2_; }
#endif
    }
    [[maybe_unused]] const unsigned long long ts_epi = CL_T();

    if (a.slab) {
        // split-K: raw partial sums, [split][position][OCp] fp32 — an accumulator quad is 4 consecutive channels of one position = one 16-byte store
        float* __restrict__ sl = a.slab + (int64_t)blockIdx.y * a.slab_m * a.OCp;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + (wm * TM + j) * 32 + l31;
#pragma unroll
            for (int i = 0; i < TOC; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int oc = oc0 + (woc * TOC + i) * 32 + 8 * q + 4 * lhi;
                    const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(sl + (int64_t)m * a.OCp + oc) = v;
                }
        }
        return;
    }
    // ---- epilogue
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const int act = a.act;
    const float slope = a.slope;
    const bool accum = a.accumulate != 0;
    if (a.coalesce) {
        // Through LDS: an accumulator register quad is 4 channels of ONE position, so direct stores are 8-byte pieces of 32 different pixels per wave
        // instruction (16 or 32 bytes of a 128- / 256-byte line at a time; round 5: the direct epilogue cost 2.4 ms of the 26 ms of gather kernels).  The
        // tile goes to LDS as [position][channel] instead and leaves as 16-byte granules in row order: a wave instruction writes 1 KB = whole lines of
        // 4-16 consecutive positions.  (Not for accumulate: the old value would have to be added after the bf16 rounding.)
        __syncthreads();                                     // every wave has left the K loop: the stage buffers are free
        uint32_t* rowoff = reinterpret_cast<uint32_t*>(smem + BM * EPITCH);
        if (tid < BM) {
            const int m = m0 + tid;
            uint32_t vo = 0xffffffffu;
            if (m < a.M && !(a.pad2 & 1)) {
                const uint32_t n = fdiv((uint32_t)m, a.div_sp);
                uint32_t r = (uint32_t)m - n * a.div_sp.div;
                const uint32_t od = fdiv(r, a.div_hw);
                r -= od * a.div_hw.div;
                const uint32_t oh = fdiv(r, a.div_w);
                const uint32_t ow = r - oh * a.div_w.div;
                vo = (uint32_t)(2 * ((int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)oh * a.y_sh + (int64_t)ow * a.y_sw + a.y_off));
            }
            rowoff[tid] = vo;
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int row = (wm * TM + j) * 32 + l31;
#pragma unroll
            for (int i = 0; i < TOC; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ocl = (woc * TOC + i) * 32 + 8 * q + 4 * lhi;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = cl_act(acc[i][j][4 * q + e], act, slope);
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    const f32x2_ p0 = {v[0], v[1]}, p1 = {v[2], v[3]};
                    u32x2 o;
                    o[0] = cl_pack2(p0[0], p0[1]);
                    o[1] = cl_pack2(p1[0], p1[1]);
                    *reinterpret_cast<u32x2*>(smem + row * EPITCH + ocl * 2) = o;
                }
        }
        __syncthreads();
        constexpr int GPR = BN / 8;                          // 16-byte granules per position row
#pragma unroll
        for (int s = 0; s < BM * GPR / NT; ++s) {
            const int idx = tid + NT * s, row = idx / GPR, c = idx % GPR;
            const uint32_t vo = rowoff[row];
            const int oc = oc0 + 8 * c;
            u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * EPITCH + c * 16);
            const uint32_t v2 = (vo != 0xffffffffu && oc < a.y_c) ? vo + (uint32_t)(2 * oc) : 0xffffffffu;
            if (accum || a.gate) {
                // dx += ... and / or the producer's activation derivative, on the row-order granules (the staged value is the bf16 rounding of the accumulator: the sum
                // round(round(acc) + old) is what adding two separately stored bf16 gradients gives)
                float f[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { f[2 * q] = cl_lo(v[q]); f[2 * q + 1] = cl_hi(v[q]); }
                if (accum) {
                    const u32x4 old = __builtin_amdgcn_raw_buffer_load_b128(yrs, v2, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { f[2 * q] += cl_lo(old[q]); f[2 * q + 1] += cl_hi(old[q]); }
                }
                if (a.gate) {
                    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cl_h*>(a.gate), 0, a.y_bytes, 0x00020000);
                    const u32x4 gv = __builtin_amdgcn_raw_buffer_load_b128(grs, v2, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (!(cl_lo(gv[q]) > 0.f)) f[2 * q] *= a.gate_slope;
                        if (!(cl_hi(gv[q]) > 0.f)) f[2 * q + 1] *= a.gate_slope;
                    }
                }
                typedef float f32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int q = 0; q < 4; ++q) { const f32x2_ t = {f[2 * q], f[2 * q + 1]}; v[q] = cl_pack2(t[0], t[1]); }
            }
            __builtin_amdgcn_raw_buffer_store_b128(v, yrs, v2, 0, 0);
        }
    } else {
    // direct: 4 consecutive output channels of one position per register quad -> one 8-byte store
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + (wm * TM + j) * 32 + l31;
        uint32_t vo = 0xffffffffu;
        if (m < a.M && !(a.pad2 & 1)) {
            const uint32_t n = fdiv((uint32_t)m, a.div_sp);
            uint32_t r = (uint32_t)m - n * a.div_sp.div;
            const uint32_t od = fdiv(r, a.div_hw);
            r -= od * a.div_hw.div;
            const uint32_t oh = fdiv(r, a.div_w);
            const uint32_t ow = r - oh * a.div_w.div;
            vo = (uint32_t)(2 * ((int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)oh * a.y_sh + (int64_t)ow * a.y_sw + a.y_off));
        }
#pragma unroll
        for (int i = 0; i < TOC; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oc = oc0 + (woc * TOC + i) * 32 + 8 * q + 4 * lhi;
                const uint32_t v2 = (oc < a.y_c && vo != 0xffffffffu) ? vo + (uint32_t)(2 * oc) : 0xffffffffu;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                if (accum) {
                    const u32x2 old = __builtin_amdgcn_raw_buffer_load_b64(yrs, v2, 0, 0);
                    v[0] += cl_lo(old[0]); v[1] += cl_hi(old[0]);
                    v[2] += cl_lo(old[1]); v[3] += cl_hi(old[1]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cl_act(v[e], act, slope);
                typedef float f32x2_ __attribute__((ext_vector_type(2)));
                const f32x2_ p0 = {v[0], v[1]}, p1 = {v[2], v[3]};
                u32x2 o;
                o[0] = cl_pack2(p0[0], p0[1]);
                o[1] = cl_pack2(p1[0], p1[1]);
                __builtin_amdgcn_raw_buffer_store_b64(o, yrs, v2, 0, 0);
            }
    }
    }
    if (a.stat) {
        // conv -> BatchNorm pairs: {sum, sum of squares} of this tile's STORED values (the bf16 roundings, as the BatchNorm op's own pass would read them), per channel.
        // Rows of padding positions and of channels past OC hold exact zeros (zero operands).  Half-wave sums by DPP, the WM waves of a channel row meet in LDS and are
        // added in a fixed order: stat[(stat_row0 + m tile)][OCp][2], every row written by exactly one workgroup.
        __syncthreads();                                     // every wave has left the K loop: the stage buffers are free
        float* sred = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < TOC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < TM; ++j) { const float v = cl_round(acc[i][j][r]); s1 += v; s2 += v * v; }
                s1 = cl_half_wave_sum(s1);
                s2 = cl_half_wave_sum(s2);
                if (l31 == 31) {
                    const int ocl = (woc * TOC + i) * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
                    sred[(wm * BN + ocl) * 2] = s1;
                    sred[(wm * BN + ocl) * 2 + 1] = s2;
                }
            }
        __syncthreads();
        float* __restrict__ dst = a.stat + ((int64_t)(a.stat_row0 + m_t) * a.OCp + oc0) * 2;
        for (int e = threadIdx.x; e < 2 * BN; e += NT) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += sred[w * BN * 2 + e];
            dst[e] = t;
        }
    }
#ifdef DCV_CL_STAMP
    if (blockIdx.x < 4096 && lane == 0) {
        const unsigned long long te = CL_T();
        unsigned long long* o = g_cl_stamps[blockIdx.x][wave];
        o[0] = ts_loop - ts_start; o[1] = ts_wait; o[2] = ts_issue; o[3] = ts_mma; o[4] = te - ts_epi; o[5] = te - ts_start; o[6] = (unsigned long long)nst; o[7] = ts_start;
    }
#endif
}
#undef CL_ISSUE

// split-K: y[position][8 channels] = act(sum over the splits, in order); one thread per 16-byte destination granule
__global__ __launch_bounds__(256) void cl_splitk_reduce_kernel(const ClGatherArgs a, int KS) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int G = a.y_c >> 3;
    if (i >= (int64_t)a.M * G) return;
    const int m = (int)(i / G), oc = (int)(i - (int64_t)m * G) * 8;
    const float* p = a.slab + (int64_t)m * a.OCp + oc;
    const int64_t stride = (int64_t)a.slab_m * a.OCp;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int k = 0; k < KS; ++k) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p + k * stride), hi = *reinterpret_cast<const f32x4*>(p + k * stride + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += lo[e]; v[4 + e] += hi[e]; }
    }
    const uint32_t n = fdiv((uint32_t)m, a.div_sp);
    uint32_t r = (uint32_t)m - n * a.div_sp.div;
    const uint32_t od = fdiv(r, a.div_hw);
    r -= od * a.div_hw.div;
    const uint32_t oh = fdiv(r, a.div_w);
    const uint32_t ow = r - oh * a.div_w.div;
    cl_h* y = a.y + ((int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)oh * a.y_sh + (int64_t)ow * a.y_sw + a.y_off) + oc;
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = cl_pack2(cl_act(v[2 * q], a.act, a.slope), cl_act(v[2 * q + 1], a.act, a.slope));
    *reinterpret_cast<u32x4*>(y) = o;
}

// Wp[cls][step][OCp][32]: thick: step = tap * cblk + cb, k = channel cb * 32 + kk;  thin: step covers taps 4 step .. 4 step + 3, k = (tap & 3) * 8 + channel
struct ClPackArgs {
    cl_h* wp[4];
    int32_t nsteps[4], T[4];
    int32_t kidx[4][64];      // filter index (kd * KH + kh) * KW + kw of tap t
    int32_t ncls, OC, OCp, C, cblk;   // cblk = 0: thin
    int32_t pad;
    int64_t ws_o, ws_r;       // w[oc * ws_o + c * ws_r + kidx]
};
__global__ __launch_bounds__(256) void cl_pack_kernel(const float* __restrict__ w, const ClPackArgs pa) {
    const int cls = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot = (int64_t)pa.nsteps[cls] * pa.OCp * 32;
    if (i >= tot) return;
    const int kk = (int)(i & 31), oc = (int)((i >> 5) % pa.OCp), step = (int)((i >> 5) / pa.OCp);
    int tap, c;
    if (pa.cblk == 0) { tap = step * 4 + (kk >> 3); c = kk & 7; }
    else { tap = step / pa.cblk; c = (step % pa.cblk) * 32 + kk; }
    float v = 0.f;
    if (oc < pa.OC && c < pa.C && tap < pa.T[cls]) v = w[(int64_t)oc * pa.ws_o + (int64_t)c * pa.ws_r + pa.kidx[cls][tap]];
    pa.wp[cls][i] = (cl_h)v;
}

// --------------------------------------------------------------------------- //
// weight gradient.  R[t][dc][gc] = sum_m D[m][dc] * G[pos(m, t)][gc]:  D = the tensor on whose position grid the sum runs ("dense": dy of a
// conv, x of a transposed conv), G the gathered one at coordinate m * stride - pad + tap.  The reduction runs over positions, so BOTH MFMA operands
// need "8 consecutive positions of one channel" per lane while memory (and the LDS image an LDS-DMA produces) is [position][channel]: the fragments
// are read with ds_read_b64_tr_b16, which transposes 4 x 16 blocks on the way out of LDS.
//   tile: 128 dense channels x 128 "virtual gathered columns" = (tap-in-tile, channel) pairs: GCB = min(GCp, 128) channels of 128 / GCB taps, so a thin
//   operand (8 padded channels) shares one tile among 16 taps and a 64-channel one among 2;
//   K step = 32 positions: D tile [32][256 B] + G tile [32][256 B], the sixteen 16-byte chunks of a row XOR-swizzled with
//   f(row) = ((row & 3) << 2) | ((row >> 2) & 3) on the DMA's source side: the transposing reads are then conflict-free;
//   D must be pixel-linear (address = m * pitch), so its per-lane DMA offsets never change and the step walks on the scalar offset; G's per-position
//   base offset and per-tap validity come from a table one small kernel fills per call (16 bytes per position);
//   grid.y splits the positions; partial tiles go to fp32 slabs summed in a fixed order (bitwise reproducible) by cl_wgrad_reduce_kernel, which also
//   scatters into the torch weight layout.
// --------------------------------------------------------------------------- //
struct ClPosEntry {
    int32_t gbase;      // byte offset of G at (n, d s - p, h s - p, w s - p), signed
    uint32_t rsv;
    uint64_t vmask;     // bit t: tap t in range for this position
};
struct ClPosArgs {
    ClPosEntry* tab;
    int32_t M, T, KH, KW;
    FastDiv div_sp, div_hw, div_w;
    int32_t s[3], p[3], k[3], gext[3];
    int64_t g_sn;
    int32_t g_sd, g_sh, g_sw, pad;
};
__global__ __launch_bounds__(256) void cl_postab_kernel(const ClPosArgs a) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= a.M) return;
    const uint32_t n = fdiv((uint32_t)m, a.div_sp);
    uint32_t r = (uint32_t)m - n * a.div_sp.div;
    const uint32_t od = fdiv(r, a.div_hw);
    r -= od * a.div_hw.div;
    const uint32_t oh = fdiv(r, a.div_w);
    const uint32_t ow = r - oh * a.div_w.div;
    const int c0[3] = {(int)od * a.s[0] - a.p[0], (int)oh * a.s[1] - a.p[1], (int)ow * a.s[2] - a.p[2]};
    uint64_t vm = 0;
    for (int t = 0; t < a.T; ++t) {
        const int kd = t / (a.KH * a.KW), kh = (t / a.KW) % a.KH, kw = t % a.KW;
        const bool ok = (unsigned)(c0[0] + kd) < (unsigned)a.gext[0] && (unsigned)(c0[1] + kh) < (unsigned)a.gext[1] && (unsigned)(c0[2] + kw) < (unsigned)a.gext[2];
        vm |= (uint64_t)ok << t;
    }
    ClPosEntry e;
    e.gbase = (int32_t)(2 * ((int64_t)n * a.g_sn + (int64_t)c0[0] * a.g_sd + (int64_t)c0[1] * a.g_sh + (int64_t)c0[2] * a.g_sw));
    e.rsv = 0;
    e.vmask = vm;
    a.tab[m] = e;
}

struct ClWgradArgs {
    const cl_h* d;
    const cl_h* g;
    const ClPosEntry* tab;
    float* slab;                 // [split][tile][128][128]
    int32_t M, chunk;            // positions; positions per split (multiple of 32)
    int32_t d_pitch2, d_cbytes;  // dense pixel pitch in bytes; bytes of a pixel's (padded) valid channels
    int32_t g_cbytes, T;         // same for the gathered tensor; taps
    int32_t tiles_d, gblocks;    // dense-channel tiles; 128-channel blocks of the gathered tensor (1 when GCp <= 128)
    int32_t gcb8, ntpt;          // 16-byte chunks per tap in a tile (GCB / 8); taps per tile (128 / GCB), or 0 = packed columns: the taps' GCB-wide column groups back
                                 // to back across the tiles (channel counts that do not divide 128 — 96: four taps in three tiles instead of four)
    int32_t tiles, S;            // tiles_d x tiles_j; position splits
    int32_t xcd_map, wtiles;     // wtiles: workgroups per split = tiles, or ceil(tiles / 2) in the narrow form
    uint32_t d_bytes, g_bytes;
    int32_t toff[64];            // byte offset of tap t relative to a position's gbase
};

__device__ __forceinline__ int cl_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// NARROW (dense operand of at most 64 channels: the 64-channel layers' and the stems' gradients): rows 64-127 of the dense tile would be zeros and the two waves that own
// them idle work — half of the MFMAs.  The workgroup takes TWO column tiles instead (images G0, G1 beside the one dense image), every wave the dense rows 0-63:
// waves 0-1 the two column halves of G0, waves 2-3 of G1.  Slab tiles keep their [128][128] shape (rows 64-127 unwritten and never read back).
template <bool NARROW>
__global__ __launch_bounds__(256, 2) void cl_wgrad_kernel(const ClWgradArgs a) {
    constexpr int IMG = 32 * 256, NG = NARROW ? 2 : 1, STAGE = (1 + NG) * IMG;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = NARROW ? 0 : wave >> 1, wj = wave & 1, sub = NARROW ? wave >> 1 : 0;
    // workgroup -> (tile, position split): ids 8 apart run on one XCD.  The (split, tile) pairs in split-major order are cut into 8 contiguous, equally long
    // ranges, one per XCD: the workgroups an XCD runs at any time are then the tiles of one or a few position splits — they read the same dense rows and
    // overlapping gathered rows, so its L2 serves every tile after the first — and every XCD gets the same number of workgroups whatever the split count
    // (first form, splits dealt out whole: 5 of 8 XCDs idle at 3 splits, 2:1 imbalance at 12).  DCV_CL_WGRAD_FLAT: tile-major ids, round-robin (A/B)
    int tile_id, split;
    if (a.xcd_map) {
        const unsigned W = (unsigned)a.wtiles * (unsigned)a.S, q = W >> 3, r = W & 7u;
        const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        const unsigned item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        tile_id = (int)(item % (unsigned)a.wtiles);
        split = (int)(item / (unsigned)a.wtiles);
    } else {
        tile_id = (int)(blockIdx.x % (unsigned)a.wtiles);
        split = (int)(blockIdx.x / (unsigned)a.wtiles);
    }
    if (split >= a.S) return;
    // (narrow: tiles_d = 1 and the work tile is the pair of column tiles 2 tile_id, 2 tile_id + 1)
    const int d_t = NARROW ? 0 : tile_id % a.tiles_d;
    int tgs[NG], gbs[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int j_t = NARROW ? 2 * tile_id + k : tile_id / a.tiles_d;
        tgs[k] = j_t / a.gblocks; gbs[k] = j_t - tgs[k] * a.gblocks;
    }
    const int m_begin = split * a.chunk;
    const int m_end = min(a.M, m_begin + a.chunk);
    const int nst = m_end > m_begin ? (m_end - m_begin + 31) / 32 : 0;

    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cl_h*>(a.d), 0, a.d_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cl_h*>(a.g), 0, a.g_bytes, 0x00020000);

    // staging roles: granule g = tid + 256 s -> row g >> 4 (position of the step), physical chunk g & 15, logical chunk ^ f(row)
    uint32_t dvo[2];
    int32_t gadd[NG][2];
    uint32_t gtap[NG][2];
    int rowi[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int gq = tid + 256 * s, row = gq >> 4, q = (gq & 15) ^ cl_swz(row);
        rowi[s] = row;
        const int dcb = (d_t * 128 + q * 8) * 2;
        dvo[s] = dcb < a.d_cbytes ? (uint32_t)(row * a.d_pitch2 + dcb) : 0xffffffffu;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            int tl, cc, tap;
            if (a.ntpt) { tl = q / a.gcb8; cc = q - tl * a.gcb8; tap = tgs[k] * a.ntpt + tl; }
            else { const int gq2 = tgs[k] * 16 + q; tap = gq2 / a.gcb8; cc = gq2 - tap * a.gcb8; tl = 0; }      // packed columns (gblocks = 1: tgs = the column tile)
            const int gcbyte = (gbs[k] * 128 + cc * 8) * 2;
            const bool ok = (a.ntpt == 0 || tl < a.ntpt) && tap < a.T && gcbyte < a.g_cbytes;      // (a pair's second tile past the last one: tap >= T)
            gtap[k][s] = ok ? (uint32_t)tap : 64u;
            gadd[k][s] = ok ? a.toff[tap & 63] + gcbyte : 0;
        }
    }

#if defined(__HIP_DEVICE_COMPILE__)
#define CL_WG_ISSUE(ST, BUF, E0, E1)                                                                                          \
    {                                                                                                                         \
        const int mrow_ = m_begin + (ST) * 32;                                                                                \
        char* db_ = smem + (BUF) * STAGE;                                                                                     \
        char* gb_ = db_ + IMG;                                                                                                \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                       \
            const uint32_t dv_ = (mrow_ + rowi[s] < m_end) ? dvo[s] : 0xffffffffu;                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void_t*)(db_ + (s * 256 + wave * 64) * 16), 16, dv_, mrow_ * a.d_pitch2, 0, 0); \
            const ClPosEntry& e_ = s ? (E1) : (E0);                                                                           \
            _Pragma("unroll") for (int k = 0; k < NG; ++k) {                                                                  \
                const bool ok_ = gtap[k][s] < 64u && ((e_.vmask >> gtap[k][s]) & 1ull) && (mrow_ + rowi[s] < m_end);          \
                const uint32_t gv_ = ok_ ? (uint32_t)(e_.gbase + gadd[k][s]) : 0xffffffffu;                                   \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (lds_void_t*)(gb_ + k * IMG + (s * 256 + wave * 64) * 16), 16, gv_, 0, 0, 0); \
            }                                                                                                                 \
        }                                                                                                                     \
    }
#else
#define CL_WG_ISSUE(ST, BUF, E0, E1) { (void)drs; (void)grs; (void)dvo; (void)gadd; (void)gtap; (void)rowi; }
#endif
#define CL_WG_ENTRY(ST, S) a.tab[min(m_begin + (ST) * 32 + rowi[S], a.M - 1)]

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing fragment reads: lane group g = lane >> 4 (columns 16 (g & 1) .., k half g >> 1), lane (qq, p) of the group supplies the address of
    // row 16 kk + 8 (g >> 1) + 4 s + qq, columns 4 p .. 4 p + 3 of its 16
    uint32_t fa[2][2][2], fb[2][2][2];   // [tile][kk][s]
    {
        const int g4 = lane >> 4, idx = lane & 15, qq = idx >> 2, p = idx & 3;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int row = kk * 16 + 8 * (g4 >> 1) + 4 * s + qq;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int cha = (wd * 64 + i * 32) / 8 + 2 * (g4 & 1) + (p >> 1);
                    const int chb = (wj * 64 + i * 32) / 8 + 2 * (g4 & 1) + (p >> 1);
                    fa[i][kk][s] = (uint32_t)(row * 256 + ((cha ^ cl_swz(row)) << 4) + 8 * (p & 1));
                    fb[i][kk][s] = (uint32_t)(IMG + sub * IMG + row * 256 + ((chb ^ cl_swz(row)) << 4) + 8 * (p & 1));
                }
            }
    }

    if (nst > 0) {
        ClPosEntry e0 = CL_WG_ENTRY(0, 0), e1 = CL_WG_ENTRY(0, 1);
        CL_WG_ISSUE(0, 0, e0, e1)
        if (nst > 1) { e0 = CL_WG_ENTRY(1, 0); e1 = CL_WG_ENTRY(1, 1); }
        for (int st = 0; st < nst; ++st) {
            const int buf = st & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (st + 1 < nst) CL_WG_ISSUE(st + 1, buf ^ 1, e0, e1)
            ClPosEntry n0 = e0, n1 = e1;
            if (st + 2 < nst) { n0 = CL_WG_ENTRY(st + 2, 0); n1 = CL_WG_ENTRY(st + 2, 1); }
            const char* sb = smem + buf * STAGE;
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                cl_h8 af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sb + fa[i][kk][0]));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sb + fa[i][kk][1]));
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sb + fb[i][kk][0]));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sb + fb[i][kk][1]));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    const s16x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    af[i] = __builtin_bit_cast(cl_h8, av);
                    bfr[i] = __builtin_bit_cast(cl_h8, bv);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = CL_MFMA(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            e0 = n0; e1 = n1;
        }
    }
    const int otile = NARROW ? 2 * tile_id + sub : tile_id;
    if (otile >= a.tiles) return;      // (the odd tile count's last pair; after the last barrier)
    float* __restrict__ out = a.slab + ((int64_t)split * a.tiles + otile) * (128 * 128);
    const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dc = wd * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int vc = wj * 64 + j * 32 + l31;
                out[dc * 128 + vc] = acc[i][j][r];
            }
}
#undef CL_WG_ISSUE
#undef CL_WG_ENTRY

// dw[dc * ws_d + gc * T + t] = sum_s slab[s][tile(dc, t, gc)][dc % 128][vcol]   (fixed order)
struct ClWgradReduceArgs {
    const float* slab;
    float* dw;
    int32_t S, tiles, tiles_d, gblocks, gcb, ntpt, T, DC, GC, lpe;
    int64_t ws_d;
    int32_t accumulate, pad;     // dw += sum (dcv_cl_conv_backward_weight_acc)
};
__global__ __launch_bounds__(256) void cl_wgrad_reduce_kernel(const ClWgradReduceArgs a) {
    // LPE lanes per output element (1, or 64 when there are many position splits: a thread walking 1000+ slabs 64 KB apart is a serial chain of
    // cache misses); fixed order either way: lane l sums slabs l, l + LPE, ... and the lanes meet in a fixed xor tree
    const int lpe = a.lpe;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = lpe == 1 ? gid : gid >> 6;
    const int l = lpe == 1 ? 0 : (int)(threadIdx.x & 63);
    const int64_t tot = (int64_t)a.DC * a.GC * a.T;
    if (i >= tot) return;
    const int t = (int)(i % a.T), gc = (int)((i / a.T) % a.GC), dc = (int)(i / ((int64_t)a.T * a.GC));
    const int d_t = dc >> 7;
    int j_t, vc;
    if (a.ntpt) { const int tg = t / a.ntpt, tl = t - tg * a.ntpt; j_t = tg * a.gblocks + (gc >> 7); vc = tl * a.gcb + (gc & 127); }
    else { const int v = t * a.gcb + gc; j_t = v >> 7; vc = v & 127; }      // packed columns
    const int tile = j_t * a.tiles_d + d_t;
    const float* p = a.slab + ((int64_t)tile * 128 + (dc & 127)) * 128 + vc;
    float s = 0.f;
    for (int k = l; k < a.S; k += lpe) s += p[(int64_t)k * a.tiles * (128 * 128)];
    if (lpe > 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (l != 0) return;
    }
    float* o = a.dw + (int64_t)dc * a.ws_d + (int64_t)gc * a.T + t;
    *o = a.accumulate ? *o + s : s;
}

// The same reduction walked in SLAB order: a workgroup owns EL consecutive slab elements and GR groups of slabs (EL x GR = 256 threads); a thread sums slabs g, g + GR, ... of its
// element (consecutive lanes read consecutive floats: whole 64- or 256-byte runs instead of one dword per lane 64 KB apart), the groups meet in LDS and are added in a fixed
// order, and the element is written once to its place in the torch-layout gradient.  Fixed order throughout: bitwise reproducible.
template <int EL, int GR>
__global__ __launch_bounds__(256) void cl_wgrad_reduce_slab_kernel(const ClWgradReduceArgs a) {
    static_assert(EL * GR == 256, "256 threads");
    __shared__ float part[GR][EL];
    const int el = threadIdx.x % EL, g = threadIdx.x / EL;
    const int64_t nel = (int64_t)a.tiles * (128 * 128);
    const int64_t e = (int64_t)blockIdx.x * EL + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < nel) {
        const float* p = a.slab + e;
        int k = g;
        for (; k + 3 * GR < a.S; k += 4 * GR) {      // four loads in flight, each into its own fixed partial sum
            s0 += p[(int64_t)k * nel]; s1 += p[(int64_t)(k + GR) * nel]; s2 += p[(int64_t)(k + 2 * GR) * nel]; s3 += p[(int64_t)(k + 3 * GR) * nel];
        }
        for (; k < a.S; k += GR) s0 += p[(int64_t)k * nel];
    }
    part[g][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g != 0 || e >= nel) return;
    float s = part[0][el];
#pragma unroll
    for (int q = 1; q < GR; ++q) s += part[q][el];
    // slab element -> (dense channel, gathered channel, tap): the inverse of cl_wgrad_reduce_kernel's addressing
    const int tile = (int)(e >> 14), r = (int)((e >> 7) & 127), vc = (int)(e & 127);
    const int j_t = tile / a.tiles_d, d_t = tile - j_t * a.tiles_d;
    const int dc = d_t * 128 + r;
    int gc, t;
    if (a.ntpt) {
        const int tg = j_t / a.gblocks, gbk = j_t - tg * a.gblocks;
        const int tl = vc / a.gcb, gcl = vc - tl * a.gcb;
        gc = gbk * 128 + gcl; t = tg * a.ntpt + tl;
        if (tl >= a.ntpt) return;
    } else { const int v = j_t * 128 + vc; t = v / a.gcb; gc = v - t * a.gcb; }      // packed columns
    if (dc >= a.DC || gc >= a.GC || t >= a.T) return;
    float* o = a.dw + (int64_t)dc * a.ws_d + (int64_t)gc * a.T + t;
    *o = a.accumulate ? *o + s : s;
}

struct ClPackThinArgs {
    cl_h* wp;
    int32_t nsteps, OCg, OCgp, C, OC, T;   // K steps (C / 32 blocks), T * OC GEMM columns, padded, source channels, real output channels, taps
    int64_t ws_o, ws_r;
};
__global__ __launch_bounds__(256) void cl_pack_thin_kernel(const float* __restrict__ w, const ClPackThinArgs pa) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot = (int64_t)pa.nsteps * pa.OCgp * 32;
    if (i >= tot) return;
    const int kk = (int)(i & 31), col = (int)((i >> 5) % pa.OCgp), step = (int)((i >> 5) / pa.OCgp);
    const int c = step * 32 + kk;
    float v = 0.f;
    if (col < pa.OCg && c < pa.C) {
        const int t = col / pa.OC, oc = col - t * pa.OC;
        v = w[(int64_t)oc * pa.ws_o + (int64_t)c * pa.ws_r + t];
    }
    pa.wp[i] = (cl_h)v;
}

struct ClCol2imArgs {
    const cl_h* z;
    cl_h* y;
    int32_t OC, zpitch, T, scatter;          // scatter: src = (dst + p - k) / s when divisible; else src = dst * s - p + k
    int32_t k[3], s[3], p[3], sext[3];       // filter, stride, padding, source extents
    int32_t dext[3], act;
    FastDiv div_sp, div_hw, div_w;           // destination pixel decode
    int64_t y_sn; int32_t y_sd, y_sh, y_sw;  // destination strides (elements)
    int32_t s_sp, s_hw, s_w, accumulate;     // source pixel linearisation (Z is contiguous): n * s_sp + d * s_hw + h * s_w + w
    float slope; int32_t pad;
    int64_t total;
};
__global__ __launch_bounds__(256) void cl_col2im_kernel(const ClCol2imArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const uint32_t n = fdiv((uint32_t)i, a.div_sp);
    uint32_t r = (uint32_t)i - n * a.div_sp.div;
    const uint32_t od = fdiv(r, a.div_hw);
    r -= od * a.div_hw.div;
    const uint32_t oh = fdiv(r, a.div_w);
    const uint32_t ow = r - oh * a.div_w.div;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // per dim: the taps whose source coordinate exists.  scatter (src = (dst + p - k) / s when divisible): k = k0 + j s with k0 = (dst + p) mod s, src = i0 - j;
    // direct (src = dst s - p + k): k = j, src = i0 + j.  One division per dim and thread, none in the loops.
    const int o[3] = {(int)od, (int)oh, (int)ow};
    int k0[3], i0[3], kst[3], ist[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (a.scatter) { const int q = o[d] + a.p[d]; k0[d] = q % a.s[d]; i0[d] = q / a.s[d]; kst[d] = a.s[d]; ist[d] = -1; }
        else { k0[d] = 0; i0[d] = o[d] * a.s[d] - a.p[d]; kst[d] = 1; ist[d] = 1; }
    }
    const cl_h* zn = a.z + (int64_t)n * a.s_sp * a.zpitch;
    for (int kd = k0[0], id = i0[0]; kd < a.k[0]; kd += kst[0], id += ist[0]) {
        if ((unsigned)id >= (unsigned)a.sext[0]) continue;
        for (int kh = k0[1], ih = i0[1]; kh < a.k[1]; kh += kst[1], ih += ist[1]) {
            if ((unsigned)ih >= (unsigned)a.sext[1]) continue;
            const cl_h* zr = zn + ((int64_t)id * a.s_hw + (int64_t)ih * a.s_w) * a.zpitch + ((kd * a.k[1] + kh) * a.k[2]) * a.OC;
            for (int kw = k0[2], iw = i0[2]; kw < a.k[2]; kw += kst[2], iw += ist[2]) {
                if ((unsigned)iw >= (unsigned)a.sext[2]) continue;
                const cl_h* zp = zr + (int64_t)iw * a.zpitch + kw * a.OC;
                for (int e = 0; e < a.OC; ++e) acc[e] += (float)zp[e];
            }
        }
    }
    cl_h* yp = a.y + (int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)oh * a.y_sh + (int64_t)ow * a.y_sw;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = e < a.OC ? acc[e] : 0.f;
        if (a.accumulate && e < a.OC) t += (float)yp[e];
        v[e] = e < a.OC ? cl_act(t, a.act, a.slope) : 0.f;
    }
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    u32x4 o4;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x2_ f = {v[2 * q], v[2 * q + 1]}; o4[q] = cl_pack2(f[0], f[1]); }
    *reinterpret_cast<u32x4*>(yp) = o4;
}


// --------------------------------------------------------------------------- //
// Thin destination, fused (round 5): the 3x3 / stride 1 / pad 1 scatter-form ops into <= 3 channels on 64-wide rows — the colour generator's RGB head forward
// (128 -> 3, tanh; generator.py:273-276) and its stem's data gradient (64 -> 1; generator.py:204-211).  The GEMM + col2im form above writes and re-reads a Z tensor
// of 32 columns per SOURCE pixel and stages the source through the tiled gather kernel: 0.93 ms for the 1.68 GB source of the RGB head.  Here the source is read
// ONCE, in whole pixels, and nothing but the thin result is written:
//   a workgroup owns a band of 16 destination rows of one image; its four waves each take 32 consecutive source pixels (half a row) per round: the pixels'
//   channels arrive by LDS-DMA as whole 16-byte-granule runs (a wave instruction = 1 KB = 4-16 whole pixels; the granules of a pixel XOR-swizzled on the source
//   side so that the "32 pixels x one granule" fragment reads are conflict-free), three units in flight per wave, no workgroup barrier on that path;
//   Z[pixel][tap, oc] = X W comes out of the MFMAs (weights held in registers: the packed thin format of cl_pack_thin_kernel) and goes to a ring of six Z rows in
//   LDS (fp32, zero columns left and right); after one barrier per round two destination rows gather their 9 x OC values from the ring, apply the activation and
//   leave as 16-byte pixels.  Bands of one image sit on one XCD (ids 8 apart), so the two halo rows a band re-reads come from its L2.
// Roofline: HBM (source once + destination once).
// --------------------------------------------------------------------------- //
struct ClThin3Args {
    const cl_h* x; cl_h* y; const cl_h* wp;
    int32_t N, H, OC, act;
    int32_t pad0, bands;            // bands of 16 rows per image
    float slope; int32_t pad;
    int64_t x_sn, y_sn;             // element strides
    int32_t x_sh, x_sw, y_sh, y_sw;
    uint32_t x_bytes, y_bytes;
};
template <int CPX>      // 16-byte granules per source pixel = C / 8: 4, 8 or 16
__global__ __launch_bounds__(256) void cl_thin3x3_kernel(const ClThin3Args a) {
    constexpr int W = 64, UNIT = 32 * CPX * 16, NBUF = 3, PPU = CPX / 2, STEPS = CPX / 4;
    constexpr int ZP = 36, ZROW = (W + 2) * ZP, RING = 6;            // floats
    constexpr int SH = CPX == 16 ? 0 : (CPX == 8 ? 1 : 2), MASK = CPX == 16 ? 15 : CPX - 1;
    __shared__ __attribute__((aligned(16))) char smem[4 * NBUF * UNIT + RING * ZROW * 4];
    float* zring = reinterpret_cast<float*>(smem + 4 * NBUF * UNIT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    // workgroup -> (image, band): the bands of an image take consecutive slots of one XCD
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const int img = (int)((slot / (unsigned)a.bands) * 8u + xcd), band = (int)(slot % (unsigned)a.bands);
    if (img >= a.N) return;
    const int b0 = band * 16, zr0 = b0 - 1;

    for (int i = tid; i < RING * ZROW; i += 256) zring[i] = 0.f;      // zero columns (and rows nobody has written yet)

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cl_h*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    // weights: A fragments of all K steps in registers
    cl_h8 wfrag[STEPS][2];
#pragma unroll
    for (int st = 0; st < STEPS; ++st)
#pragma unroll
        for (int h = 0; h < 2; ++h) wfrag[st][h] = *reinterpret_cast<const cl_h8*>(a.wp + ((st * 32 + l31) * 32 + (2 * h + lhi) * 8));
    // staging roles: granule gi = i * 64 + lane of the unit -> pixel gi / CPX, physical granule gi % CPX holding logical granule ^ swz(pixel)
    uint32_t voff[PPU];
#pragma unroll
    for (int i = 0; i < PPU; ++i) {
        const int gi = i * 64 + lane, px = gi / CPX, pg = gi % CPX;
        voff[i] = (uint32_t)(px * a.x_sw * 2 + ((pg ^ ((px >> SH) & MASK)) << 4));
    }
    char* xbuf = smem + wave * (NBUF * UNIT);
    const int half = wave & 1, wrow = wave >> 1;
    const int64_t img_base = (int64_t)img * a.x_sn * 2;

#if defined(__HIP_DEVICE_COMPILE__)
#define T3_ISSUE(ROUND, BUF)                                                                                                   \
    {                                                                                                                          \
        const int zr_ = zr0 + 2 * (ROUND) + wrow;                                                                              \
        const bool ok_ = zr_ >= 0 && zr_ < a.H;                                                                                \
        const uint32_t so_ = ok_ ? (uint32_t)(img_base + ((int64_t)zr_ * a.x_sh + (int64_t)half * 32 * a.x_sw) * 2) : 0u;     \
        _Pragma("unroll") for (int i = 0; i < PPU; ++i)                                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(xbuf + (BUF) * UNIT + i * 1024), 16, ok_ ? voff[i] : 0xffffffffu, so_, 0, 0); \
    }
#else
#define T3_ISSUE(ROUND, BUF) { (void)xrs; (void)voff; (void)img_base; }
#endif
    constexpr int ROUNDS = 9;      // Z rows b0 - 1 .. b0 + 16
    T3_ISSUE(0, 0)
    T3_ISSUE(1, 1)
    __syncthreads();               // ring zeroed
    int buf = 0;
    for (int t = 0; t < ROUNDS; ++t) {
        if (t + 2 < ROUNDS) { T3_ISSUE(t + 2, (buf + 2) % NBUF) cl_wait_vm<2 * PPU>(); }
        else if (t + 1 < ROUNDS) cl_wait_vm<PPU>();
        else cl_wait_vm<0>();
        // Z of this wave's 32 pixels: 32 columns x 32 pixels
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const char* xb = xbuf + buf * UNIT;
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int lg = st * 4 + 2 * h + lhi;
                const cl_h8 b8 = *reinterpret_cast<const cl_h8*>(xb + l31 * (CPX * 16) + ((lg ^ ((l31 >> SH) & MASK)) << 4));
                acc = CL_MFMA(wfrag[st][h], b8, acc, 0, 0, 0);
            }
        {
            const int zr = zr0 + 2 * t + wrow;
            float* zp = zring + ((zr - zr0) % RING) * ZROW + (1 + half * 32 + l31) * ZP;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                *reinterpret_cast<f32x4*>(zp + 8 * q + 4 * lhi) = v;
            }
        }
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        __syncthreads();
        // destination rows zr0 + 2 t - 1 and zr0 + 2 t: out[o][c][oc] = sum_{dr, dc} Z[o + dr][c + dc][((1 - dr) * 3 + (1 - dc)) * OC + oc]
        if (tid < 128) {
            const int o = zr0 + 2 * t - 1 + (tid >> 6), c = tid & 63;
            if (o >= b0 && o < b0 + 16 && o < a.H) {
                float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dr = -1; dr <= 1; ++dr) {
                    const float* zrow = zring + ((o + dr - zr0) % RING) * ZROW + (1 + c) * ZP;
#pragma unroll
                    for (int dc = -1; dc <= 1; ++dc) {
                        const float* zq = zrow + dc * ZP + ((1 - dr) * 3 + (1 - dc)) * a.OC;
#pragma unroll
                        for (int oc = 0; oc < 3; ++oc) r[oc] += oc < a.OC ? zq[oc] : 0.f;
                    }
                }
                const uint32_t vo = (uint32_t)(2 * ((int64_t)img * a.y_sn + (int64_t)o * a.y_sh + (int64_t)c * a.y_sw));
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = e < a.OC ? cl_act(r[e], a.act, a.slope) : 0.f;
                typedef float f32x2_ __attribute__((ext_vector_type(2)));
                u32x4 o4;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const f32x2_ f = {v[2 * q], v[2 * q + 1]}; o4[q] = cl_pack2(f[0], f[1]); }
                __builtin_amdgcn_raw_buffer_store_b128(o4, yrs, vo, 0, 0);
            }
        }
    }
#undef T3_ISSUE
}


// --------------------------------------------------------------------------- //
// Thin source, fused (round 5): the 3x3 / stride 1 / pad 1 direct-form ops out of <= 8 channels on 64-wide rows — the colour generator's stem forward (1 or 2 -> 64,
// LeakyReLU; generator.py:204-211) and the RGB head's data gradient (3 -> 128; generator.py:273-276).  These are bound by their OUTPUT (0.84 / 1.68 GB): with K = (tap, 8
// channels) an MFMA operand fragment is exactly one 16-byte source pixel, so the band's 18 source rows (1 KB each, one LDS-DMA apiece, a zero pixel left and right) are
// staged ONCE and every fragment read is that image at a per-tap offset; a wave then owns 32 consecutive destination pixels: 6 MFMAs per 32 output channels, the
// activation, a wave-private LDS transpose, and the pixels leave as 16-byte granules in memory order (32 pixels x OC channels: one contiguous run when the tensor
// owns its pixels).  No K loop, no workgroup barrier after the staging.  Roofline: HBM (destination once).
// --------------------------------------------------------------------------- //
struct ClWiden3Args {
    const cl_h* x; cl_h* y; const cl_h* wp;      // wp: the thin packed format [3 steps][OCp][32], k = (tap & 3) * 8 + channel
    int32_t N, H, OCp, act;
    int32_t bands, pad0;
    float slope; int32_t pad1;
    int64_t x_sn, y_sn;
    int32_t x_sh, x_sw, y_sh, y_sw;
    uint32_t x_bytes, y_bytes;
};
template <int OCB>      // 32-channel blocks of the destination (OC = 32 OCB)
__global__ __launch_bounds__(256) void cl_widen3x3_kernel(const ClWiden3Args a) {
    constexpr int W = 64, RP = (W + 2) * 16, ROWS = 18, EP = OCB * 64 + 16, GPR = OCB * 4;
    __shared__ __attribute__((aligned(16))) char smem[ROWS * RP + 4 * 32 * EP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const int img = (int)((slot / (unsigned)a.bands) * 8u + xcd), band = (int)(slot % (unsigned)a.bands);
    if (img >= a.N) return;
    const int b0 = band * 16;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cl_h*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    if (tid < 2 * ROWS) *reinterpret_cast<u32x4*>(smem + (tid >> 1) * RP + (tid & 1) * (W + 1) * 16) = u32x4{0u, 0u, 0u, 0u};      // the zero pixels left and right of each row
#if defined(__HIP_DEVICE_COMPILE__)
    for (int r = wave; r < ROWS; r += 4) {      // source rows b0 - 1 .. b0 + 16; rows outside the image arrive as zeros (out-of-range offset)
        const int sr = b0 - 1 + r;
        const bool ok = sr >= 0 && sr < a.H;
        const uint32_t so = ok ? (uint32_t)(((int64_t)img * a.x_sn + (int64_t)sr * a.x_sh) * 2) : 0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(smem + r * RP + 16), 16, ok ? (uint32_t)(lane * a.x_sw * 2) : 0xffffffffu, so, 0, 0);
    }
#endif
    // weights: A fragments of the three K steps (12 tap slots, 9 used; the packed buffer holds zeros in the others) for every 32-channel block
    cl_h8 wfrag[OCB][3][2];
#pragma unroll
    for (int ob = 0; ob < OCB; ++ob)
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int h = 0; h < 2; ++h) wfrag[ob][st][h] = *reinterpret_cast<const cl_h8*>(a.wp + (((int64_t)st * a.OCp + ob * 32 + l31) * 32 + (2 * h + lhi) * 8));
    // B fragment of (step, half): the source pixel of tap = 4 step + 2 half + lhi at this lane's destination pixel; tap slots past the ninth read the zero pixel
    uint32_t boff[3][2];
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tap = st * 4 + 2 * h + lhi;
            boff[st][h] = tap < 9 ? (uint32_t)(((tap / 3) * (W + 2) + l31 + tap % 3) * 16) : 0xffffffffu;
        }
    cl_wait_vm<0>();
    __syncthreads();
    char* tw = smem + ROWS * RP + wave * (32 * EP);
    const int act = a.act;
    const float slope = a.slope;
    for (int u = wave; u < 32; u += 4) {
        const int r = u >> 1, half = u & 1;
        const uint32_t ubase = (uint32_t)((r * (W + 2) + half * 32) * 16);
        cl_h8 b8[3][2];
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int h = 0; h < 2; ++h) b8[st][h] = *reinterpret_cast<const cl_h8*>(smem + (boff[st][h] == 0xffffffffu ? 0u : ubase + boff[st][h]));
#pragma unroll
        for (int ob = 0; ob < OCB; ++ob) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int st = 0; st < 3; ++st)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc = CL_MFMA(wfrag[ob][st][h], b8[st][h], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cl_act(acc[4 * q + e], act, slope);
                typedef float f32x2_ __attribute__((ext_vector_type(2)));
                const f32x2_ p0 = {v[0], v[1]}, p1 = {v[2], v[3]};
                u32x2 o;
                o[0] = cl_pack2(p0[0], p0[1]);
                o[1] = cl_pack2(p1[0], p1[1]);
                *reinterpret_cast<u32x2*>(tw + l31 * EP + (ob * 32 + 8 * q + 4 * lhi) * 2) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes, before its lanes read one another's (LDS operations of a wave complete in order)
        const uint32_t ybase = (uint32_t)(2 * ((int64_t)img * a.y_sn + (int64_t)(b0 + r) * a.y_sh + (int64_t)(half * 32) * a.y_sw));
#pragma unroll
        for (int it = 0; it < GPR / 2; ++it) {
            const int idx = it * 64 + lane, px = idx / GPR, c = idx % GPR;
            const u32x4 v = *reinterpret_cast<const u32x4*>(tw + px * EP + c * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, yrs, ybase + (uint32_t)(px * a.y_sw * 2 + c * 16), 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next unit overwrites the transpose image
    }
}


// --------------------------------------------------------------------------- //
// Thin source, fused, 3-D (round 5): the video / gradient discriminators' stems — Conv3d(<= 8 -> 32, 4x4x4, stride (1, 2, 2), padding (0, 1, 1)) + LeakyReLU on
// 64 x 64 frames (discriminator.py:181-186,288-291).  The tiled thin gather reads 64 taps x 16 bytes per output position out of L2 (1.4 GB per launch for a 105 MB
// input: 0.18 ms, nine launches per iteration).  Here a workgroup owns 4 output rows of one (sample, output depth): the 10 x 4 input rows it needs (a 1 KB LDS-DMA
// each, a zero pixel left and right) are staged once, a wave owns one output row of 32 pixels, an MFMA operand fragment is ONE staged pixel at a compile-time
// offset per (tap, half) — no per-lane address arithmetic in the K loop — and the 32 x 32 result leaves through a wave-private LDS transpose as 16-byte granules.
// The 13 output depths of a row band run back to back on one XCD (they share three of their four input planes).  Roofline: HBM (input + output once).
// --------------------------------------------------------------------------- //
struct ClStem3Args {
    const cl_h* x; cl_h* y; const cl_h* wp;      // wp: thin packed format [16 steps][32][32], step = kd * 4 + kh, k = kw * 8 + channel
    int32_t N, OD, act, pad0;
    float slope; int32_t pad1;
    int64_t x_sn, y_sn;
    int32_t x_sd, x_sh, x_sw, y_sd, y_sh, y_sw;
    uint32_t x_bytes, y_bytes;
};
__global__ __launch_bounds__(256) void cl_stem3d_kernel(const ClStem3Args a) {
    constexpr int W = 64, RP = (W + 2) * 16, NR = 10, ROWS = 4 * NR, EP = 64 + 16;      // 4 depth planes x 10 input rows; transpose image: 32 pixels x (32 channels + pad)
    __shared__ __attribute__((aligned(16))) char smem[ROWS * RP + 4 * 32 * EP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    // workgroup -> (sample, row band, output depth): ids 8 apart on one XCD; the output depths of a band are consecutive slots
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const int od = (int)(slot % (unsigned)a.OD);
    const unsigned nb = (slot / (unsigned)a.OD) * 8u + xcd;      // (sample, band) pair
    const int n = (int)(nb >> 3), band = (int)(nb & 7u);
    if (n >= a.N) return;
    const int oh0 = band * 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cl_h*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    if (tid < 2 * ROWS) *reinterpret_cast<u32x4*>(smem + (tid >> 1) * RP + (tid & 1) * (W + 1) * 16) = u32x4{0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
    for (int r = wave; r < ROWS; r += 4) {      // LDS row r = (depth tap r / 10, input row 2 oh0 - 1 + r % 10)
        const int kd = r / NR, ih = 2 * oh0 - 1 + (r - kd * NR);
        const bool ok = ih >= 0 && ih < 64;
        const uint32_t so = ok ? (uint32_t)(((int64_t)n * a.x_sn + (int64_t)(od + kd) * a.x_sd + (int64_t)ih * a.x_sh) * 2) : 0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(smem + r * RP + 16), 16, ok ? (uint32_t)(lane * a.x_sw * 2) : 0xffffffffu, so, 0, 0);
    }
#endif
    cl_h8 wfrag[16][2];
#pragma unroll
    for (int st = 0; st < 16; ++st)
#pragma unroll
        for (int h = 0; h < 2; ++h) wfrag[st][h] = *reinterpret_cast<const cl_h8*>(a.wp + ((st * 32 + l31) * 32 + (2 * h + lhi) * 8));
    cl_wait_vm<0>();
    __syncthreads();
    // this wave's output row oh0 + wave: lane's pixel ow = l31; tap (kd, kh, kw = 2 h + lhi) sits at LDS row kd * 10 + 2 wave + kh, pixel 2 ow + kw (LDS pixel 0 = input column -1)
    const char* base = smem + (2 * wave) * RP + (2 * l31 + lhi) * 16;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int st = 0; st < 16; ++st)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const cl_h8 b8 = *reinterpret_cast<const cl_h8*>(base + ((st >> 2) * NR + (st & 3)) * RP + h * 32);
            acc = CL_MFMA(wfrag[st][h], b8, acc, 0, 0, 0);
        }
    char* tw = smem + ROWS * RP + wave * (32 * EP);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u32x2 o;
        o[0] = cl_pack2(cl_act(acc[4 * q], a.act, a.slope), cl_act(acc[4 * q + 1], a.act, a.slope));
        o[1] = cl_pack2(cl_act(acc[4 * q + 2], a.act, a.slope), cl_act(acc[4 * q + 3], a.act, a.slope));
        *reinterpret_cast<u32x2*>(tw + l31 * EP + (8 * q + 4 * lhi) * 2) = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t ybase = (uint32_t)(2 * ((int64_t)n * a.y_sn + (int64_t)od * a.y_sd + (int64_t)(oh0 + wave) * a.y_sh));
#pragma unroll
    for (int it = 0; it < 2; ++it) {      // 32 pixels x 4 granules
        const int idx = it * 64 + lane, px = idx >> 2, c = idx & 3;
        const u32x4 v = *reinterpret_cast<const u32x4*>(tw + px * EP + c * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, yrs, ybase + (uint32_t)(px * a.y_sw * 2 + c * 16), 0, 0);
    }
}

// --------------------------------------------------------------------------- //
// Patch-staged scatter form, 2-D, 4x4 / stride 2 / padding 1 (round 5): ConvTranspose2d forward (the generators' up-sampling layers) and the data gradient of
// Conv2d(4, 2, 1).  The tiled gather runs the four stride-parity classes as four GEMMs: every class stages its own 2 x 2 taps of the source (each source pixel goes
// through the texture path 16 times per 64 destination channels, in half lines — tools/lds_dma_rate.hip: that path takes an LDS-DMA instruction line by line, so
// half lines halve it), and on the layers with 16-32 K steps a workgroup spends 15-40 % of its life in prologue + epilogue (cycle stamps).  Here a workgroup owns
// 256 source positions (8 rows of a 32-wide image ... 16 whole 4 x 4 images) and ALL FOUR classes of them = 1024 destination positions x 64 channels:
//   * per 32-channel block the source patch with its one-pixel halo is staged ONCE (340-576 rows of 64 bytes, double-buffered across blocks); the 16 (class, tap)
//     pairs are 9 source offsets, applied by the fragment reads (a lane's row + a per-offset constant);
//   * the weights of a block stream through a two-slot ring in four chunks of four (class, tap) pairs x 64 channels x 32 k = 16 KB of whole lines, ordered by
//     source offset: centre (4 classes) | up, down | left, right | the four corners — 9 activation fragment reads per 32 MFMAs instead of 16;
//   * 8 waves, wave = 32 source positions x 64 channels x 4 classes = 128 accumulator registers; one barrier per chunk (16 MFMAs per wave), counted vmcnt waits
//     (a chunk's weights are issued one chunk ahead, the next block's patch two to four chunks ahead);
//   * per K block the texture path moves 22-37 KB of patch + 64 KB of weights for 512 MFMAs (~40 % of its rate at full MFMA speed), the LDS ~60 %: MFMA-bound;
//   * epilogue through a wave-private LDS transpose: the two column classes of a destination row interleave into whole 128-byte pixels, 16-byte row-order stores;
//     BatchNorm sums of the stored values per (patch, channel) for conv -> BatchNorm pairs, summed over the waves in a fixed order.
// Packed weights: [oc tile][k block][chunk][pair][64 oc][32 k], granules pre-swizzled (the DMA is a linear copy).  Roofline: MFMA.
// --------------------------------------------------------------------------- //
struct ClPatchArgs {
    const cl_h* x; cl_h* y; const cl_h* wp; float* stat;
    int32_t N, H, W, C32;          // source images, extent, 32-channel K blocks
    int32_t OCp, octiles, NI, PRI; // padded destination channels (64 octiles); images and rows of an image per patch (NI PRI W = 32 NW)
    int32_t RI, prows, bands, npatch;   // patch rows per image = (PRI + 2)(W + 2); NI RI; H / PRI; patches
    int32_t wlog, plog, act, y_c;  // log2 W, log2 (PRI W); channels to store
    float slope; int32_t total;    // workgroups that have work = npatch octiles
    int64_t x_sn, y_sn;
    int32_t x_sh, x_sw, y_sh, y_sw;
    uint32_t x_bytes, y_bytes, w_bytes, pad;
};
constexpr int CLP_B = 16384;      // a weight chunk

// NW waves = 32 NW source positions per workgroup: 8 (one workgroup per CU) or 4 (two independent ones, which overlap one's staging and fragment reads with the other's MFMAs;
// each stages the weight chunks itself: 2x the weight bytes through the texture path)
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void cl_patch_convt_kernel(const ClPatchArgs a) {
    constexpr int CLP_AMAX = (NW == 8 ? 37 : 19) * 1024;      // a patch buffer: up to 576 / 288 rows in pieces of 16
    __shared__ __attribute__((aligned(16))) char smem[2 * CLP_AMAX + 2 * CLP_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    // workgroup -> (patch, oc tile): ids 8 apart (one XCD) walk a contiguous range of patches, the oc tiles of a patch next to one another
    const unsigned per = gridDim.x >> 3;
    const unsigned lin = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (lin >= (unsigned)a.total) return;
    const int patch = (int)(lin / (unsigned)a.octiles), octile = (int)(lin % (unsigned)a.octiles);
    const int img0 = (patch / a.bands) * a.NI, r0 = (patch % a.bands) * a.PRI;
    const int W2 = a.W + 2;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cl_h*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cl_h*>(a.wp), 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);

    // ---- staging roles: piece = 16 patch rows x 4 granules; wave w owns pieces w, w + NW, ... (at most 5); granule -> row g >> 2, physical chunk g & 3 holds logical chunk ^ ((row >> 2) & 3)
    const int npiece = (a.prows + 15) >> 4;
    const int np_w = (npiece - wave + NW - 1) / NW;      // pieces of this wave
    uint32_t xvo[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int g = (wave + NW * j) * 64 + lane, row = g >> 2, c = (g & 3) ^ ((row >> 2) & 3);
        uint32_t vo = 0xffffffffu;
        if (row < a.prows) {
            const int im = row / a.RI, rem = row - im * a.RI;
            const int ry = rem / W2, rx = rem - ry * W2;
            const int sy = r0 + ry - 1, sx = rx - 1, n = img0 + im;
            if (n < a.N && (unsigned)sy < (unsigned)a.H && (unsigned)sx < (unsigned)a.W)
                vo = (uint32_t)(2 * ((int64_t)n * a.x_sn + (int64_t)sy * a.x_sh + (int64_t)sx * a.x_sw) + 16 * c);
        }
        xvo[j] = vo;
    }
    // ---- wave -> (position pair pp, channel group ocg): 64 source positions (two 32-position groups) x 32 channels x 4 classes.  Per K16 slice and (class, tap) pair the
    // wave reads ONE weight fragment for two MFMAs (first form: 32 positions x 64 channels per wave = two weight fragments per activation fragment, every wave reading the
    // whole 16 KB chunk: 36 fragment reads per 16 MFMAs, the LDS array at its limit; now 17)
    const int pp = wave >> 1, ocg = wave & 1;
    // fragment addresses.  Activations: source position q = 64 pp + 32 pg + l31 -> patch row; offset (dy, dx) adds dy (W + 2) + dx rows
    uint32_t aaddr[2][9][2];
#pragma unroll
    for (int pg = 0; pg < 2; ++pg) {
        const int q = pp * 64 + pg * 32 + l31;
        const int im = q >> a.plog, rem = q & ((1 << a.plog) - 1);
        const int r = rem >> a.wlog, c = rem & (a.W - 1);
        const int row0 = im * a.RI + (r + 1) * W2 + (c + 1);
#pragma unroll
        for (int o = 0; o < 9; ++o) {
            const int row = row0 + (o / 3 - 1) * W2 + (o % 3 - 1);
#pragma unroll
            for (int k = 0; k < 2; ++k) aaddr[pg][o][k] = (uint32_t)(row * 64 + (((2 * k + lhi) ^ ((row >> 2) & 3)) << 4));
        }
    }
    // weights: row = pair * 64 + 32 ocg + l31 of the chunk: (row >> 2) & 3 = (l31 >> 2) & 3
    uint32_t boff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) boff[k] = (uint32_t)((ocg * 32 + l31) * 64 + (((2 * k + lhi) ^ ((l31 >> 2) & 3)) << 4));

    f32x16 acc[4][2];      // [class][position group]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int S = a.C32 * 4;      // chunks
    constexpr int BPW = 16 / NW;      // weight pieces per wave and chunk
    const uint32_t wbase = (uint32_t)octile * (uint32_t)a.C32 * 4u * (uint32_t)CLP_B + (uint32_t)(BPW * wave) * 1024u + (uint32_t)lane * 16u;
#if defined(__HIP_DEVICE_COMPILE__)
#define CLP_ISSUE_A(J, CB, BUF)                                                                                                       \
    if ((J) < np_w) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(smem + (BUF) * CLP_AMAX + (wave + NW * (J)) * 1024), 16, xvo[J], (CB) * 64, 0, 0);
#define CLP_ISSUE_B(STEP)                                                                                                             \
    {                                                                                                                                 \
        char* d_ = smem + 2 * CLP_AMAX + ((STEP) & 1) * CLP_B + (BPW * wave) * 1024;                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < BPW; ++i_)                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_t*)(d_ + i_ * 1024), 16, wbase + (uint32_t)i_ * 1024u, (STEP) * CLP_B, 0, 0); \
    }
#else
#define CLP_ISSUE_A(J, CB, BUF) { (void)xvo; (void)np_w; (void)xrs; }
#define CLP_ISSUE_B(STEP) { (void)wbase; (void)wrs; }
#endif
#define CLP_A(O) av0 = *reinterpret_cast<const cl_h8*>(Ab + aaddr[0][O][k]); av1 = *reinterpret_cast<const cl_h8*>(Ab + aaddr[1][O][k]);
#define CLP_CT(CT, CLS)                                                                                                               \
    {                                                                                                                                 \
        const cl_h8 b_ = *reinterpret_cast<const cl_h8*>(Bb + boff[k] + (CT) * 4096);                                                 \
        acc[CLS][0] = CL_MFMA(b_, av0, acc[CLS][0], 0, 0, 0);                                                                         \
        acc[CLS][1] = CL_MFMA(b_, av1, acc[CLS][1], 0, 0, 0);                                                                         \
    }
    // raw barrier: __syncthreads() would put s_waitcnt vmcnt(0) in front of it while LDS-DMAs are pending and drain the patch pieces the counted waits leave in flight
    // (the wave's own fragment reads are done: their MFMAs have consumed them; the memory clobbers keep the compiler from moving LDS accesses or DMAs across)
#define CLP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // what the wave may leave in flight at the top of a chunk: the patch pieces it issued AFTER the weights in the previous chunk
    auto wait_top = [&](int left) {
        if (left <= 0) cl_wait_vm<0>();
        else if (left == 1) cl_wait_vm<1>();
        else if (left == 2) cl_wait_vm<2>();
        else cl_wait_vm<3>();
    };
    CLP_ISSUE_A(0, 0, 0) CLP_ISSUE_A(1, 0, 0) CLP_ISSUE_A(2, 0, 0) CLP_ISSUE_A(3, 0, 0) CLP_ISSUE_A(4, 0, 0)
    CLP_ISSUE_B(0)
    const int n0 = np_w < 3 ? np_w : 3, n1 = np_w - n0;      // patch pieces issued in chunk 0 / chunk 1 of a block
    for (int cb = 0; cb < a.C32; ++cb) {
        const char* Ab = smem + (cb & 1) * CLP_AMAX;
        const bool nextb = cb + 1 < a.C32;
        cl_h8 av0, av1;
        // ---- chunk 0: the centre offset, all four classes
        {
            cl_wait_vm<0>();
            CLP_BARRIER();
            CLP_ISSUE_B(cb * 4 + 1)
            if (nextb) { CLP_ISSUE_A(0, cb + 1, (cb + 1) & 1) CLP_ISSUE_A(1, cb + 1, (cb + 1) & 1) CLP_ISSUE_A(2, cb + 1, (cb + 1) & 1) }
            const char* Bb = smem + 2 * CLP_AMAX;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int k = 0; k < 2; ++k) { CLP_A(4) CLP_CT(0, 0) CLP_CT(1, 1) CLP_CT(2, 2) CLP_CT(3, 3) }
            __builtin_amdgcn_s_setprio(0);
        }
        // ---- chunk 1: one row up (row classes 0), one row down (row classes 1)
        {
            wait_top(nextb ? n0 : 0);
            CLP_BARRIER();
            CLP_ISSUE_B(cb * 4 + 2)
            if (nextb) { CLP_ISSUE_A(3, cb + 1, (cb + 1) & 1) CLP_ISSUE_A(4, cb + 1, (cb + 1) & 1) }
            const char* Bb = smem + 2 * CLP_AMAX + CLP_B;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int k = 0; k < 2; ++k) { CLP_A(1) CLP_CT(0, 0) CLP_CT(1, 1) CLP_A(7) CLP_CT(2, 2) CLP_CT(3, 3) }
            __builtin_amdgcn_s_setprio(0);
        }
        // ---- chunk 2: one column left (column classes 0), one column right (column classes 1)
        {
            wait_top(nextb ? n1 : 0);
            CLP_BARRIER();
            CLP_ISSUE_B(cb * 4 + 3)
            const char* Bb = smem + 2 * CLP_AMAX;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int k = 0; k < 2; ++k) { CLP_A(3) CLP_CT(0, 0) CLP_CT(1, 2) CLP_A(5) CLP_CT(2, 1) CLP_CT(3, 3) }
            __builtin_amdgcn_s_setprio(0);
        }
        // ---- chunk 3: the corners
        {
            cl_wait_vm<0>();
            CLP_BARRIER();
            const int st = cb * 4 + 3;
            if (st + 1 < S) CLP_ISSUE_B(st + 1)
            const char* Bb = smem + 2 * CLP_AMAX + CLP_B;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int k = 0; k < 2; ++k) { CLP_A(0) CLP_CT(0, 0) CLP_A(2) CLP_CT(1, 1) CLP_A(6) CLP_CT(2, 2) CLP_A(8) CLP_CT(3, 3) }
            __builtin_amdgcn_s_setprio(0);
        }
    }
#undef CLP_BARRIER
#undef CLP_ISSUE_A
#undef CLP_ISSUE_B
#undef CLP_A
#undef CLP_CT
    __syncthreads();      // every wave has left the K loop: the staging buffers are free

    // ---- epilogue: per destination-row class py the workgroup's 64 NW destination pixels (source position p, column class px) -> [2 p + px][64 channels] in LDS (each wave
    // its 32 channels), out as 16-byte granules in pixel order: whole 128-byte pixels (for a 32-wide source row: whole destination rows, 8 KB contiguous)
    constexpr int EP = 128 + 16, NPX = 64 * NW;
    uint32_t* rowoff = reinterpret_cast<uint32_t*>(smem + NPX * EP);
    for (int j = tid; j < NPX; j += 64 * NW) {
        const int q = j >> 1, px = j & 1;
        const int im = q >> a.plog, rem = q & ((1 << a.plog) - 1);
        const int r = rem >> a.wlog, c = rem & (a.W - 1);
        const int n = img0 + im;
        rowoff[j] = n < a.N ? (uint32_t)(2 * ((int64_t)n * a.y_sn + (int64_t)(2 * (r0 + r)) * a.y_sh + (int64_t)(2 * c + px) * a.y_sw)) : 0xffffffffu;
    }
    const int act = a.act;
    const float slope = a.slope;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x16& t = acc[py * 2 + px][pg];
                    u32x2 o;
                    o[0] = cl_pack2(cl_act(t[4 * q4], act, slope), cl_act(t[4 * q4 + 1], act, slope));
                    o[1] = cl_pack2(cl_act(t[4 * q4 + 2], act, slope), cl_act(t[4 * q4 + 3], act, slope));
                    *reinterpret_cast<u32x2*>(smem + (2 * (pp * 64 + pg * 32 + l31) + px) * EP + (ocg * 32 + 8 * q4 + 4 * lhi) * 2) = o;
                }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * (64 * NW) + tid, j = idx >> 3, c = idx & 7;
            const uint32_t ro = rowoff[j];
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + j * EP + c * 16);
            const uint32_t vo = (ro != 0xffffffffu && octile * 64 + 8 * c < a.y_c) ? ro + (uint32_t)(py * a.y_sh * 2 + (octile * 64 + 8 * c) * 2) : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b128(v, yrs, vo, 0, 0);
        }
        __syncthreads();      // reads done before the next row class overwrites the image
    }
    if (a.stat) {
        // {sum, sum of squares} of the STORED values per channel over this patch's 1024 destination positions (rows of images past N hold exact zeros): half-wave sums by
        // DPP, the NW / 2 position pairs of a channel meet in LDS and are added in pair order: stat[patch][OCp][2]
        float* sred = reinterpret_cast<float*>(smem);      // (the epilogue's last barrier has passed: the image is free)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int cls = 0; cls < 4; ++cls)
#pragma unroll
                for (int pg = 0; pg < 2; ++pg) { const float v = cl_round(acc[cls][pg][r]); s1 += v; s2 += v * v; }
            s1 = cl_half_wave_sum(s1);
            s2 = cl_half_wave_sum(s2);
            if (l31 == 31) {
                const int ocl = ocg * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3);
                sred[(pp * 64 + ocl) * 2] = s1;
                sred[(pp * 64 + ocl) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW / 2; ++w) t += sred[w * 128 + tid];
            a.stat[((int64_t)patch * a.OCp + octile * 64) * 2 + tid] = t;
        }
    }
}

// weights of cl_patch_convt_kernel: [oc tile][k block][chunk][pair][64 oc][32 k]; (chunk, pair) -> (class, filter row, filter column) as the kernel's chunks use them;
// the four 16-byte granules of a row stored at position ^ ((oc >> 2) & 3)
struct ClPackPatchArgs {
    cl_h* wp;
    int32_t OC, OCp, C, C32, KW, pad;
    int64_t ws_o, ws_r;
};
__global__ __launch_bounds__(256) void cl_pack_patch_kernel(const float* __restrict__ w, const ClPackPatchArgs pa) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot = (int64_t)pa.OCp * pa.C32 * 16 * 32;
    if (i >= tot) return;
    const int kk = (int)(i & 31), ocl = (int)((i >> 5) & 63), ct = (int)((i >> 11) & 3), ch = (int)((i >> 13) & 3);
    const int64_t rest = i >> 15;
    const int cb = (int)(rest % pa.C32), octile = (int)(rest / pa.C32);
    // logical k of this physical slot
    const int gl = (kk >> 3) ^ ((ocl >> 2) & 3), k = gl * 8 + (kk & 7);
    // (chunk, pair) -> class (py, px) and source offset (dy, dx); filter index: row class 0: dy 0 -> 1, -1 -> 3; row class 1: dy 0 -> 2, +1 -> 0 (columns alike)
    int py, px, dy, dx;
    if (ch == 0) { py = ct >> 1; px = ct & 1; dy = 0; dx = 0; }
    else if (ch == 1) { py = ct >> 1; px = ct & 1; dy = py ? 1 : -1; dx = 0; }
    else if (ch == 2) { px = ct >> 1; py = ct & 1; dy = 0; dx = px ? 1 : -1; }
    else { py = ct >> 1; px = ct & 1; dy = py ? 1 : -1; dx = px ? 1 : -1; }
    const int ky = dy == 0 ? (py ? 2 : 1) : (py ? 0 : 3), kx = dx == 0 ? (px ? 2 : 1) : (px ? 0 : 3);
    const int oc = octile * 64 + ocl, c = cb * 32 + k;
    float v = 0.f;
    if (oc < pa.OC && c < pa.C) v = w[(int64_t)oc * pa.ws_o + (int64_t)c * pa.ws_r + ky * pa.KW + kx];
    pa.wp[i] = (cl_h)v;
}

struct ClTile { int bn, bm; };
// (oc, positions) tile of a destination with OC channels.  Two more tiles exist and are OFF by default (DCV_CL_TILES: bit 0 = 128 x 256, bit 1 = 96-wide), both
// measured neutral in round 5 (profiles/r05_ab_cl16.txt, call 7: layer table 32.6 / 32.7 / 32.6 / 32.8 ms, iteration 41.7-42.2 ms for all four settings):
// 128 x 256 on 8 waves (512 threads) issues 3 LDS-DMA instructions per wave and 8 MFMAs instead of 4 — worth 0-7 % per layer, the 8-wave barrier takes it back;
// 96 x 256 (channel counts that are multiples of 96 but not of 128: the geometry generator at ngf 96) saves the padded quarter of a 128-wide tile but runs at 168
// registers = two workgroups per CU.
static int cl_tile_opts() {
    static const int v = getenv("DCV_CL_TILES") ? atoi(getenv("DCV_CL_TILES")) : 0;
    return v;
}
static ClTile cl_pick_tile(int OC) {
    if ((cl_tile_opts() & 2) && OC > 64 && OC % 96 == 0 && OC % 128 != 0) return {96, 256};
    if (OC > 64) return (cl_tile_opts() & 1) ? ClTile{128, 256} : ClTile{128, 128};
    // (64 x 128 since the end of round 5: the layer table is the same with either — 29.80 / 29.85 ms — the iteration with its three lanes 0.2 ms shorter, 3 of 3 pairs;
    //  bit 2 of DCV_CL_TILES brings 64 x 256 back)
    if (OC > 32) return (cl_tile_opts() & 4) ? ClTile{64, 256} : ClTile{64, 128};
    return {32, 256};
}

static size_t cl_class_pack_bytes(const ClClass& c, int C, int OC) {
    const ClTile tc = cl_pick_tile(OC);
    const int OCp = (OC + tc.bn - 1) / tc.bn * tc.bn;
    const int T = c.t[0].n * c.t[1].n * c.t[2].n;
    if (T == 0) return 0;
    const int nsteps = cl_thin(C) ? (T + 3) / 4 : T * (cl_cp(C) / 32);
    return align_up((size_t)nsteps * OCp * 64, 256);
}

// src (gathered, RC channels) -> dst (OC channels); weight element (oc, rc, tap) at oc * ws_o + rc * ws_r + kidx
struct ClPlan {
    std::vector<ClClass> cls;
    int RC, OC;
    int64_t ws_o, ws_r;
    int KH, KW;
};

static int cl_make_plan(int which, const dcv_conv_geom* g, const dcv_dims5* xd, const dcv_dims5* yd, ClPlan* pl) {
    const int k[3] = {g->kd, g->kh, g->kw}, s[3] = {g->sd, g->sh, g->sw}, p[3] = {g->pd, g->ph, g->pw};
    const int xi[3] = {xd->d, xd->h, xd->w}, yo[3] = {yd->d, yd->h, yd->w};
    const int T = k[0] * k[1] * k[2];
    if (k[0] > 4 || k[1] > 4 || k[2] > 4) return fail(DCV_EUNSUPPORTED, "cl conv: filters up to 4 taps per dim");
    for (int d = 0; d < 3; ++d) {
        const int want = g->transposed ? (xi[d] - 1) * s[d] - 2 * p[d] + k[d] : (xi[d] + 2 * p[d] - k[d]) / s[d] + 1;
        if (want != yo[d]) return fail(DCV_EINVAL, "cl conv: output extent %d along dim %d, geometry gives %d", yo[d], d, want);
    }
    if (xd->c != g->cin || yd->c != g->cout || xd->n != yd->n) return fail(DCV_EINVAL, "cl conv: channel / batch mismatch");
    const bool direct = (which == 0 && !g->transposed) || (which == 1 && g->transposed);
    const dcv_dims5& src = (which == 0) ? *xd : *yd;
    const dcv_dims5& dst = (which == 0) ? *yd : *xd;
    pl->RC = src.c; pl->OC = dst.c; pl->KH = k[1]; pl->KW = k[2];
    if (direct) {
        pl->cls = cl_direct_classes(k, s, p, (which == 0) ? yo : xi, (which == 0) ? xi : yo);
        pl->ws_o = (int64_t)pl->RC * T; pl->ws_r = T;
    } else {
        pl->cls = cl_scatter_classes(k, s, p, (which == 0) ? yo : xi, (which == 0) ? xi : yo);
        pl->ws_o = T; pl->ws_r = (int64_t)pl->OC * T;
    }
    return DCV_OK;
}

// thin destination fed by a wide source: the GEMM + col2im form (above)
static bool cl_thin_out(const ClPlan& pl, const dcv_conv_geom* g) {
    const int T = g->kd * g->kh * g->kw;
    static const bool off = getenv("DCV_CL_NO_COL2IM") != nullptr;
    // (up to 64 GEMM columns: beyond that — the 4x4x4 data gradient into the 3-channel video, 192 columns — the gather form measured faster, 0.46 vs 1.12 ms)
    return pl.OC <= 8 && !cl_thin(pl.RC) && T * pl.OC <= 64 && !off;
}
static inline int cl_pitch(int c) { return c <= 8 ? 8 : (c + 31) / 32 * 32; }

// cl_patch_convt_kernel's plan.  Decided on SHAPES only (the packed weights carry its layout behind the tiled gather's, so a call the run-time conditions exclude —
// accumulate, gate, a misaligned view — still has the other form's tiles)
struct ClPatchPlan { bool ok; int NP, NI, PRI, RI, prows, bands, npatch, wlog, plog, C32, OCp, octiles; size_t pack_bytes; };
static ClPatchPlan cl_patch_plan(int which, const dcv_conv_geom* g, const ClPlan& pl, const dcv_dims5& src, const dcv_dims5& dst) {
    ClPatchPlan r;
    memset(&r, 0, sizeof(r));
    static const bool off = getenv("DCV_CL_NO_PATCH") != nullptr;      // A/B only
    const bool direct = (which == 0 && !g->transposed) || (which == 1 && g->transposed);
    if (off || direct || g->kd != 1 || g->kh != 4 || g->kw != 4 || g->sd != 1 || g->sh != 2 || g->sw != 2 || g->pd != 0 || g->ph != 1 || g->pw != 1) return r;
    if (src.d != 1 || dst.d != 1 || dst.h != 2 * src.h || dst.w != 2 * src.w) return r;
    const int W = src.w, H = src.h;
    if ((W != 4 && W != 8 && W != 16 && W != 32) || H < 1 || (H & (H - 1))) return r;
    if (pl.RC < 32 || pl.RC % 32 || pl.OC < 64 || pl.OC % 8) return r;      // (destination channels past the last whole 64: zero weights, never stored)
    static const int np_env = getenv("DCV_CL_PATCH_NP") ? atoi(getenv("DCV_CL_PATCH_NP")) : 0;      // A/B only: 256 = eight waves, one workgroup per CU
    const int NP = np_env == 256 ? 256 : 128;
    r.NP = NP;
    if (H * W >= NP) { r.NI = 1; r.PRI = NP / W; if (r.PRI < 1 || H % r.PRI) return r; }
    else { r.NI = NP / (H * W); r.PRI = H; }
    r.RI = (r.PRI + 2) * (W + 2); r.prows = r.NI * r.RI;
    if (r.prows > (NP == 256 ? 576 : 288)) return r;
    r.bands = H / r.PRI;
    r.npatch = (src.n + r.NI - 1) / r.NI * r.bands;
    for (r.wlog = 0; (1 << r.wlog) < W; ++r.wlog) {}
    for (r.plog = 0; (1 << r.plog) < r.PRI * W; ++r.plog) {}
    r.C32 = pl.RC / 32; r.OCp = (pl.OC + 63) / 64 * 64; r.octiles = r.OCp / 64;
    r.pack_bytes = (size_t)r.octiles * r.C32 * 4 * CLP_B;
    r.ok = true;
    return r;
}
static size_t cl_generic_pack_bytes(const ClPlan& pl) {
    size_t tot = 0;
    for (const ClClass& c : pl.cls) tot += cl_class_pack_bytes(c, pl.RC, pl.OC);
    return align_up(tot, 256);
}

static int cl_check_tensor(const dcv_dims5& d, const char* tag) {
    if (d.c > 1 && d.sc != 1) return fail(DCV_EINVAL, "%s: channels-last tensor expected (channel stride 1, got %lld)", tag, (long long)d.sc);
    if ((d.w > 1 && d.sw % 8) || (d.h > 1 && d.sh % 8) || (d.d > 1 && d.sd % 8) || (d.n > 1 && d.sn % 8)) return fail(DCV_EINVAL, "%s: pixel pitch must be a multiple of 8 elements", tag);
    return DCV_OK;
}
// bytes from the tensor's first element to one past its last pixel's padded channels (all strides non-negative)
static int64_t cl_extent_bytes(const dcv_dims5& d, int cpad) {
    return 2 * ((int64_t)(d.n - 1) * d.sn + (int64_t)(d.d - 1) * d.sd + (int64_t)(d.h - 1) * d.sh + (int64_t)(d.w - 1) * d.sw + cpad);
}

static void cl_launch_tile(const ClTile tc, const ClGatherPack& pk, bool thin, dim3 grid, hipStream_t s);
// (Round 5, measured and not instantiated: NS = 3 and 4 — two / three K steps in flight behind counted vmcnt waits, at 3 or 2 workgroups per CU instead of 4 —
// ran the surreal-depth1 layer table in 39.5 / 41.3 ms against 37.9 and the iteration in 49.3 / 51.4 ms against 47.2 (profiles/r05_ab_cl16.txt): what bounds the
// loop is the issue cost of the LDS-DMA instructions, 4-5 per wave and K step against 8 MFMAs, not the distance of the prefetch; fewer waves per SIMD lose more.)
// split-K plan of a single-class, thick, plain (no statistics / accumulate / gate) call whose output has at most 4 positions per SAMPLE (the latent layers: 1x1 and 2x2
// outputs, M = 1-4 x batch positions for K loops of 128-384 steps).  The decision and the split count depend on the geometry only, never on the batch: every output
// element's K sum keeps one fixed order whatever the batch size (tests/test_cl16_oracle_gpu.py::test_cl16_batch_split_identity_b100 holds B = 100 against B = 16 bit for
// bit; a tile-count rule, tried first, broke exactly that).  KS = 0: not split.
struct ClSplitK { int KS, ks_per; size_t slab_bytes; int slab_m; };
static ClSplitK cl_splitk_plan(int ncls, bool thin, int64_t M, int per_sample, int OCp, int nsteps, const ClTile tc, int ocs) {
    ClSplitK r = {0, 0, 0, 0};
    static const bool off = getenv("DCV_CL_NO_SPLITK") != nullptr;      // A/B only
    if (off || ncls != 1 || thin || ocs % 8 || per_sample > 4 || nsteps < 64) return r;
    const int KS = std::min(8, nsteps / 8);
    const int64_t tiles_m = (M + tc.bm - 1) / tc.bm;
    r.ks_per = (nsteps + KS - 1) / KS;
    r.KS = (nsteps + r.ks_per - 1) / r.ks_per;
    r.slab_m = (int)(tiles_m * tc.bm);
    r.slab_bytes = (size_t)r.KS * r.slab_m * OCp * sizeof(float);
    return r;
}
template <int TOC, int TM, int WOC, int WM>
static void cl_launch_gather(const ClGatherPack& pk, bool thin, dim3 grid, hipStream_t s) {
    if (thin) hipLaunchKernelGGL((cl_gather_kernel<TOC, TM, WOC, WM, true, 2>), grid, dim3(64 * WOC * WM), 0, s, pk);
    else hipLaunchKernelGGL((cl_gather_kernel<TOC, TM, WOC, WM, false, 2>), grid, dim3(64 * WOC * WM), 0, s, pk);
}
static void cl_launch_tile(const ClTile tc, const ClGatherPack& pk, bool thin, dim3 grid, hipStream_t s) {
    if (tc.bn == 128 && tc.bm == 256) cl_launch_gather<2, 2, 2, 4>(pk, thin, grid, s);
    else if (tc.bn == 128) cl_launch_gather<2, 2, 2, 2>(pk, thin, grid, s);
    else if (tc.bn == 96) cl_launch_gather<3, 2, 1, 4>(pk, thin, grid, s);
    else if (tc.bn == 64 && tc.bm == 128) cl_launch_gather<2, 1, 1, 4>(pk, thin, grid, s);
    else if (tc.bn == 64) cl_launch_gather<2, 2, 1, 4>(pk, thin, grid, s);
    else cl_launch_gather<1, 2, 1, 4>(pk, thin, grid, s);
}


// The position table of a weight-gradient call depends on the geometry and the two tensors' shapes / strides only: built once per (device, key) and kept
// (16 bytes per dense position; the iteration's ~30 distinct layers hold ~0.5 GB at B = 100).  DCV_CL_NO_POSTAB_CACHE=1: rebuild into the workspace per call.
struct ClTabKey {
    int dev; int32_t v[24]; int64_t w[4];
    bool operator<(const ClTabKey& o) const { return memcmp(this, &o, sizeof(*this)) < 0; }
};
static std::mutex g_tab_mu;
static std::map<ClTabKey, ClPosEntry*> g_tabs;
static constexpr size_t CL_TAB_CACHE_MAX = 256;

static bool cl_pixel_linear(const dcv_dims5& d, int64_t* pitch) {
    int64_t p = 0;
    if (d.w > 1) p = d.sw; else if (d.h > 1) p = d.sh; else if (d.d > 1) p = d.sd; else if (d.n > 1) p = d.sn; else p = pad8(d.c);
    if ((d.h > 1 && d.sh != (int64_t)d.w * p) || (d.d > 1 && d.sd != (int64_t)d.h * d.w * p) || (d.n > 1 && d.sn != (int64_t)d.d * d.h * d.w * p)) return false;
    *pitch = p;
    return true;
}
struct ClWgradPlan {
    int64_t M;
    int DC, GC, T, tiles_d, gblocks, gcb, ntpt, tiles_j, tiles, S, chunk, wtiles;
    bool narrow;
    size_t tab_bytes, slab_bytes;
};
static int cl_wgrad_plan(const dcv_conv_geom* g, const dcv_dims5& D, const dcv_dims5& G, ClWgradPlan* p) {
    p->M = (int64_t)D.n * D.d * D.h * D.w;
    if (p->M >= (1ll << 31) || p->M == 0) return fail(DCV_EUNSUPPORTED, "cl wgrad: position count");
    p->DC = D.c; p->GC = G.c; p->T = g->kd * g->kh * g->kw;
    if (p->T > 64) return fail(DCV_EUNSUPPORTED, "cl wgrad: more than 64 taps");
    const int gcp = pad8(G.c);
    if (gcp >= 128) { p->gcb = 128; p->gblocks = (gcp + 127) / 128; p->ntpt = 1; }
    else { p->gcb = gcp; p->gblocks = 1; p->ntpt = 128 / gcp; }
    p->tiles_d = (pad8(D.c) + 127) / 128;
    p->tiles_j = (p->T + p->ntpt - 1) / p->ntpt * p->gblocks;
    static const bool no_packed = getenv("DCV_CL_WGRAD_NO_PACKED") != nullptr;      // A/B only
    if (!no_packed && gcp < 128 && 128 % gcp != 0) {      // packed columns: a tile's 128 columns run across tap boundaries (cl_wgrad_kernel, ntpt = 0)
        const int tj = (p->T * gcp + 127) / 128;
        if (tj < p->tiles_j) { p->ntpt = 0; p->tiles_j = tj; }
    }
    p->tiles = p->tiles_d * p->tiles_j;
    // narrow form (cl_wgrad_kernel<true>): a dense operand of <= 64 channels fills half of a 128-row tile: column tiles in pairs instead
    static const bool no_narrow = getenv("DCV_CL_WGRAD_NO_NARROW") != nullptr;      // A/B only
    p->narrow = !no_narrow && pad8(D.c) <= 64 && p->tiles_j >= 2;
    p->wtiles = p->narrow ? (p->tiles_j + 1) / 2 : p->tiles;
    // position splits: ~768 workgroups (three per CU) however few tiles the op has, at least 16 K steps (512 positions) each; measured over the
    // surreal-depth1 layer table with the first reduce kernel: 13.8 ms of weight gradients per iteration at 1024, 15.9 at 2048, 20.1 at 4096 (slab traffic); with the
    // slab-order reduce, whole iterations on one box: 48.6 / 48.2 / 48.9 / 48.6 ms at 640 / 768 / 896 / 1024 (surreal-depth1), 41.8 / 41.6 / 42.1 / 42.1 (isogd-depth)
    static const int64_t target = getenv("DCV_CL_WGRAD_WGS") ? atoll(getenv("DCV_CL_WGRAD_WGS")) : 768;
    int64_t S = (target + p->wtiles - 1) / p->wtiles;
    const int64_t maxS = std::max<int64_t>(1, p->M / 512);
    S = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(S, maxS), 2048));
    int64_t chunk = ((p->M + S - 1) / S + 31) / 32 * 32;
    S = (p->M + chunk - 1) / chunk;
    p->S = (int)S; p->chunk = (int)chunk;
    p->tab_bytes = align_up((size_t)p->M * sizeof(ClPosEntry), 256);
    p->slab_bytes = (size_t)S * p->tiles * 128 * 128 * sizeof(float);
    return DCV_OK;
}

#ifdef DCV_CL_FP16
}  // inline namespace clf16
#endif
}  // namespace dcv

using namespace dcv;

extern "C" {

#ifdef DCV_CL_STAMP
int dcv_cl_debug_read_stamps(unsigned long long* host, int zero) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cl_stamps), sizeof(g_cl_stamps)) != hipSuccess) return -1;
    if (zero) { static unsigned long long z[8]; (void)z; hipMemset(nullptr, 0, 0); void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_cl_stamps)) == hipSuccess) (void)hipMemset(p, 0, sizeof(g_cl_stamps)); }
    return 0;
}
#endif

size_t dcv_cl_packed_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which) {
    ClPlan pl;
    if (!g || !x || !y || (which != 0 && which != 1) || cl_make_plan(which, g, x, y, &pl) != DCV_OK) return 0;
    if (cl_thin_out(pl, g)) {
        const int OCg = g->kd * g->kh * g->kw * pl.OC;
        const ClTile tc = cl_pick_tile(OCg);
        return align_up((size_t)(cl_cp(pl.RC) / 32) * ((OCg + tc.bn - 1) / tc.bn * tc.bn) * 64, 256) + 256;
    }
    // (+ the patch-staged form's tiles behind the tiled gather's, where the shapes allow that kernel: cl_patch_plan)
    const ClPatchPlan pp = cl_patch_plan(which, g, pl, which == 0 ? *x : *y, which == 0 ? *y : *x);
    return cl_generic_pack_bytes(pl) + (pp.ok ? pp.pack_bytes : 0) + 256;
}

// scratch a forward / backward-data call needs (the Z tensor of the thin-destination form; 0 otherwise)
size_t dcv_cl_conv_workspace_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which) {
    ClPlan pl;
    if (!g || !x || !y || (which != 0 && which != 1) || cl_make_plan(which, g, x, y, &pl) != DCV_OK) return 0;
    if (!cl_thin_out(pl, g)) {
        // split-K slabs of a single-class call with few position tiles (cl_splitk_plan; the run splits only if this much workspace is there)
        size_t slab = 0;
        if (pl.cls.size() == 1) {
            const dcv_dims5& dst = (which == 0) ? *y : *x;
            const ClClass& c = pl.cls[0];
            const ClTile tc = cl_pick_tile(pl.OC);
            const int OCp = (pl.OC + tc.bn - 1) / tc.bn * tc.bn, T = c.t[0].n * c.t[1].n * c.t[2].n;
            const int nsteps = cl_thin(pl.RC) ? (T + 3) / 4 : T * (cl_cp(pl.RC) / 32);
            slab = cl_splitk_plan(1, cl_thin(pl.RC), (int64_t)dst.n * c.o_ext[0] * c.o_ext[1] * c.o_ext[2], c.o_ext[0] * c.o_ext[1] * c.o_ext[2], OCp, nsteps, tc, 8).slab_bytes;
        }
        return 256 + slab;
    }
    const dcv_dims5& src = (which == 0) ? *x : *y;
    return (size_t)src.n * src.d * src.h * src.w * cl_pitch(g->kd * g->kh * g->kw * pl.OC) * 2 + 512;
}

int dcv_cl_pack_weights(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y, int which, const float* w, void* packed, size_t bytes, void* stream) {
    ClPlan pl;
    if (!g || !x || !y || !w || !packed) return fail(DCV_EINVAL, "cl_pack_weights: null pointer");
    int rc = cl_make_plan(which, g, x, y, &pl);
    if (rc != DCV_OK) return rc;
    if (cl_thin_out(pl, g)) {
        ClPackThinArgs ta;
        memset(&ta, 0, sizeof(ta));
        ta.T = g->kd * g->kh * g->kw; ta.OC = pl.OC; ta.OCg = ta.T * pl.OC; ta.C = pl.RC;
        const ClTile tg = cl_pick_tile(ta.OCg);
        ta.OCgp = (ta.OCg + tg.bn - 1) / tg.bn * tg.bn;
        ta.nsteps = cl_cp(pl.RC) / 32;
        ta.ws_o = pl.ws_o; ta.ws_r = pl.ws_r;
        ta.wp = static_cast<cl_h*>(packed);
        const int64_t tot = (int64_t)ta.nsteps * ta.OCgp * 32;
        if ((size_t)tot * 2 > bytes) return fail(DCV_EWORKSPACE, "cl_pack_weights: buffer too small");
        hipLaunchKernelGGL(cl_pack_thin_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, ta);
        DCV_LAUNCH_CHECK();
        return DCV_OK;
    }
    const ClTile tc = cl_pick_tile(pl.OC);
    ClPackArgs pa;
    memset(&pa, 0, sizeof(pa));
    size_t off = 0;
    int n = 0;
    int64_t maxtot = 0;
    pa.OC = pl.OC; pa.OCp = (pl.OC + tc.bn - 1) / tc.bn * tc.bn; pa.C = pl.RC; pa.cblk = cl_thin(pl.RC) ? 0 : cl_cp(pl.RC) / 32;
    pa.ws_o = pl.ws_o; pa.ws_r = pl.ws_r;
    for (const ClClass& c : pl.cls) {
        const int T = c.t[0].n * c.t[1].n * c.t[2].n;
        const size_t b = cl_class_pack_bytes(c, pl.RC, pl.OC);
        if (T == 0) continue;
        if (off + b > bytes) return fail(DCV_EWORKSPACE, "cl_pack_weights: buffer too small");
        pa.wp[n] = reinterpret_cast<cl_h*>(static_cast<char*>(packed) + off);
        pa.T[n] = T;
        pa.nsteps[n] = cl_thin(pl.RC) ? (T + 3) / 4 : T * (cl_cp(pl.RC) / 32);
        for (int t = 0; t < T; ++t) {
            const int ud = t / (c.t[1].n * c.t[2].n), uh = (t / c.t[2].n) % c.t[1].n, uw = t % c.t[2].n;
            pa.kidx[n][t] = (c.t[0].kidx[ud] * pl.KH + c.t[1].kidx[uh]) * pl.KW + c.t[2].kidx[uw];
        }
        maxtot = std::max<int64_t>(maxtot, (int64_t)pa.nsteps[n] * pa.OCp * 32);
        off += b;
        if (++n > 4) return fail(DCV_EUNSUPPORTED, "cl_pack_weights: more than 4 position classes");
    }
    if (n == 0) return DCV_OK;
    pa.ncls = n;
    hipLaunchKernelGGL(cl_pack_kernel, dim3((unsigned)((maxtot + 255) / 256), (unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream), w, pa);
    DCV_LAUNCH_CHECK();
    const ClPatchPlan pp = cl_patch_plan(which, g, pl, which == 0 ? *x : *y, which == 0 ? *y : *x);
    if (pp.ok) {
        const size_t o2 = cl_generic_pack_bytes(pl);
        if (o2 + pp.pack_bytes > bytes) return fail(DCV_EWORKSPACE, "cl_pack_weights: buffer too small");
        ClPackPatchArgs pq;
        memset(&pq, 0, sizeof(pq));
        pq.wp = reinterpret_cast<cl_h*>(static_cast<char*>(packed) + o2);
        pq.OC = pl.OC; pq.OCp = pp.OCp; pq.C = pl.RC; pq.C32 = pp.C32; pq.KW = pl.KW; pq.ws_o = pl.ws_o; pq.ws_r = pl.ws_r;
        const int64_t tot = (int64_t)pp.OCp * pp.C32 * 16 * 32;
        hipLaunchKernelGGL(cl_pack_patch_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, pq);
        DCV_LAUNCH_CHECK();
    }
    return DCV_OK;
}

// which: 0 forward (x -> y), 1 backward-data (dy -> dx); src / dst are bf16 channels-last
static int cl_conv_thin_out(int which, const dcv_conv_geom* g, const ClPlan& pl, const void* src_p, const dcv_dims5& src, const void* packed, void* dst_p,
                            const dcv_dims5& dst, int act, float slope, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    const int T = g->kd * g->kh * g->kw, OCg = T * pl.OC, zp = cl_pitch(OCg), Cp = cl_cp(pl.RC);
    const int64_t Msrc = (int64_t)src.n * src.d * src.h * src.w;
    if (Msrc >= (1ll << 31)) return fail(DCV_EUNSUPPORTED, "cl conv: too many source pixels");
    const size_t zbytes = (size_t)Msrc * zp * 2;
    if (!ws || ws_bytes < zbytes) return fail(DCV_EWORKSPACE, "cl conv: workspace too small for the thin-destination form (%zu needed, %zu given)", zbytes, ws_bytes);
    if (src.w > 1 && src.sw < Cp) return fail(DCV_EINVAL, "cl conv: source pixel pitch %lld < padded channels %d", (long long)src.sw, Cp);
    if (dst.w > 1 && dst.sw < 8) return fail(DCV_EINVAL, "cl conv: thin destination needs a pixel pitch of 8");
    const int64_t xb = cl_extent_bytes(src, Cp);
    if (xb >= (1ll << 31) || zbytes >= (1ull << 31)) return fail(DCV_EUNSUPPORTED, "cl conv: tensors beyond 2 GB");
    {
        // fused form (cl_thin3x3_kernel): 2-D 3x3 / stride 1 / pad 1 scatter-form, 64-wide rows, rows in bands of 16, 32 / 64 / 128 source channels, <= 3 destination channels
        static const bool no_fused = getenv("DCV_CL_NO_THIN3") != nullptr;      // A/B only
        const bool direct3 = (which == 0 && !g->transposed) || (which == 1 && g->transposed);
        const int C3 = pl.RC;
        if (!no_fused && !accumulate && !direct3 && g->kd == 1 && g->kh == 3 && g->kw == 3 && g->sd == 1 && g->sh == 1 && g->sw == 1 && g->pd == 0 && g->ph == 1 && g->pw == 1 &&
            src.d == 1 && dst.d == 1 && src.w == 64 && src.h % 16 == 0 && dst.w == 64 && dst.h == src.h && pl.OC <= 3 && (C3 == 32 || C3 == 64 || C3 == 128) &&
            (reinterpret_cast<uintptr_t>(src_p) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst_p) & 15) == 0 && dst.sw >= 8 && src.sw >= C3) {
            const int64_t xb3 = cl_extent_bytes(src, C3), yb3 = cl_extent_bytes(dst, 8);
            if (xb3 < (1ll << 31) && yb3 < (1ll << 31)) {
                ClThin3Args t;
                memset(&t, 0, sizeof(t));
                t.x = static_cast<const cl_h*>(src_p); t.y = static_cast<cl_h*>(dst_p); t.wp = static_cast<const cl_h*>(packed);
                t.N = src.n; t.H = src.h; t.OC = pl.OC; t.act = act; t.bands = src.h / 16; t.slope = slope;
                t.x_sn = src.sn; t.y_sn = dst.sn; t.x_sh = (int32_t)src.sh; t.x_sw = (int32_t)src.sw; t.y_sh = (int32_t)dst.sh; t.y_sw = (int32_t)dst.sw;
                t.x_bytes = (uint32_t)xb3; t.y_bytes = (uint32_t)yb3;
                const unsigned nwg = (unsigned)((src.n + 7) / 8 * 8 * t.bands);
                if (C3 == 128) hipLaunchKernelGGL((cl_thin3x3_kernel<16>), dim3(nwg), dim3(256), 0, st, t);
                else if (C3 == 64) hipLaunchKernelGGL((cl_thin3x3_kernel<8>), dim3(nwg), dim3(256), 0, st, t);
                else hipLaunchKernelGGL((cl_thin3x3_kernel<4>), dim3(nwg), dim3(256), 0, st, t);
                snprintf(g_last_kernel, sizeof(g_last_kernel), "cl_thin3x3_kernel<%d> (fused GEMM + tap gather, thin destination, " CL_HALF_NAME " channels-last)", C3 / 8);
                DCV_LAUNCH_CHECK();
                return DCV_OK;
            }
        }
    }
    const ClTile tc = cl_pick_tile(OCg);
    const int OCgp = (OCg + tc.bn - 1) / tc.bn * tc.bn;
    // (1) Z = X (1x1) Wg : every source pixel once
    ClGatherPack pk;
    memset(&pk, 0, sizeof(pk));
    ClGatherArgs& a = pk.c[0];
    a.x = static_cast<const cl_h*>(src_p); a.y = static_cast<cl_h*>(ws); a.wp = static_cast<const cl_h*>(packed);
    a.M = (int)Msrc; a.OCp = OCgp; a.T = 1; a.cblk = Cp / 32; a.nsteps = Cp / 32; a.y_c = pad8(OCg); a.x_cmax = 2 * pad8(pl.RC);      // Z owns its pixels (pitch zp >= pad8): whole 8-column groups, zeros past OCg
    a.div_sp = make_fastdiv((uint32_t)(src.d * src.h * src.w)); a.div_hw = make_fastdiv((uint32_t)(src.h * src.w)); a.div_w = make_fastdiv((uint32_t)src.w);
    ClDim one;
    memset(&one, 0, sizeof(one));
    one.n = 1; one.mul = 1; one.base = 0;
    a.td = one; a.td.size = src.d; a.th = one; a.th.size = src.h; a.tw = one; a.tw.size = src.w;
    a.x_sn = src.sn; a.x_sd = (int32_t)src.sd; a.x_sh = (int32_t)src.sh; a.x_sw = (int32_t)src.sw;
    a.y_sn = (int64_t)src.d * src.h * src.w * zp; a.y_sd = src.h * src.w * zp; a.y_sh = src.w * zp; a.y_sw = zp; a.y_off = 0;
    a.act = DCV_ACT_NONE; a.x_bytes = (uint32_t)xb; a.y_bytes = (uint32_t)zbytes;
    a.coalesce = (a.y_c % 8 == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 && !getenv("DCV_CL_DIRECT_EPILOGUE")) ? 1 : 0;
    for (int i = 1; i < 4; ++i) pk.c[i] = pk.c[0];
    pk.ncls = 1; pk.tiles_oc = OCgp / tc.bn; pk.tiles_m = (int)((Msrc + tc.bm - 1) / tc.bm);
    const dim3 grid((unsigned)((pk.tiles_m + 7) / 8 * 8 * pk.tiles_oc));
    cl_launch_tile(tc, pk, false, grid, st);
    DCV_LAUNCH_CHECK();
    // (2) gather the taps of every destination pixel
    ClCol2imArgs c;
    memset(&c, 0, sizeof(c));
    c.z = static_cast<const cl_h*>(ws); c.y = static_cast<cl_h*>(dst_p);
    c.OC = pl.OC; c.zpitch = zp; c.T = T;
    const bool direct = (which == 0 && !g->transposed) || (which == 1 && g->transposed);
    c.scatter = direct ? 0 : 1;
    c.k[0] = g->kd; c.k[1] = g->kh; c.k[2] = g->kw; c.s[0] = g->sd; c.s[1] = g->sh; c.s[2] = g->sw; c.p[0] = g->pd; c.p[1] = g->ph; c.p[2] = g->pw;
    c.sext[0] = src.d; c.sext[1] = src.h; c.sext[2] = src.w; c.dext[0] = dst.d; c.dext[1] = dst.h; c.dext[2] = dst.w;
    c.act = act; c.slope = slope; c.accumulate = accumulate;
    c.div_sp = make_fastdiv((uint32_t)(dst.d * dst.h * dst.w)); c.div_hw = make_fastdiv((uint32_t)(dst.h * dst.w)); c.div_w = make_fastdiv((uint32_t)dst.w);
    c.y_sn = dst.sn; c.y_sd = (int32_t)dst.sd; c.y_sh = (int32_t)dst.sh; c.y_sw = (int32_t)dst.sw;
    c.s_sp = src.d * src.h * src.w; c.s_hw = src.h * src.w; c.s_w = src.w;
    c.total = (int64_t)dst.n * dst.d * dst.h * dst.w;
    if (c.total >= (1ll << 31)) return fail(DCV_EUNSUPPORTED, "cl conv: too many destination pixels");
    hipLaunchKernelGGL(cl_col2im_kernel, dim3((unsigned)((c.total + 255) / 256)), dim3(256), 0, st, c);
    snprintf(g_last_kernel, sizeof(g_last_kernel), "cl_gather_kernel<%d x %d tile> as a 1x1 GEMM over the source + cl_col2im_kernel (thin destination, " CL_HALF_NAME " channels-last)", tc.bn, tc.bm);
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

// `stat` (forward, no activation, no accumulation, a full-channel destination): per-tile BatchNorm sums from the epilogue, *nparts rows of *pitch channels x {sum, sum^2};
// *nparts stays 0 where the form does not produce them (thin destinations) and the caller's BatchNorm op takes its own statistics.
static int cl_conv_run(int which, const dcv_conv_geom* g, const void* src_p, const dcv_dims5* xd, const void* packed, void* dst_p, const dcv_dims5* yd,
                       int act, float slope, int accumulate, void* ws, size_t ws_bytes, void* stream,
                       float* stat = nullptr, size_t stat_bytes = 0, int* nparts = nullptr, int* pitch = nullptr, const void* gate = nullptr, float gate_slope = 0.f) {
    if (nparts) *nparts = 0;
    if (pitch) *pitch = 0;
    if (!g || !xd || !yd || !src_p || !packed || !dst_p) return fail(DCV_EINVAL, "cl conv: null pointer");
    ClPlan pl;
    int rc = cl_make_plan(which, g, xd, yd, &pl);
    if (rc != DCV_OK) return rc;
    const dcv_dims5& src = (which == 0) ? *xd : *yd;
    const dcv_dims5& dst = (which == 0) ? *yd : *xd;
    if ((rc = cl_check_tensor(src, "cl conv source")) != DCV_OK || (rc = cl_check_tensor(dst, "cl conv destination")) != DCV_OK) return rc;
    if (gate && (cl_thin_out(pl, g) || cl_thin(pl.RC))) return fail(DCV_EUNSUPPORTED, "cl conv: no gated epilogue in the thin forms");
    if (cl_thin_out(pl, g)) return cl_conv_thin_out(which, g, pl, src_p, src, packed, dst_p, dst, act, slope, accumulate, ws, ws_bytes, static_cast<hipStream_t>(stream));
    const bool thin = cl_thin(pl.RC);
    const int Cp = cl_cp(pl.RC);
    // the gathered tensor's channel slice must be readable in whole K granules: pixel pitch >= padded channel count
    if (src.w > 1 && src.sw < (thin ? 8 : Cp)) return fail(DCV_EINVAL, "cl conv: source pixel pitch %lld < padded channels %d", (long long)src.sw, thin ? 8 : Cp);
    // stores are 4-channel groups; a destination that owns its whole pixel (not a channel slice of a wider buffer) gets all pad8(OC) channels written —
    // zeros past OC — so nobody has to clear a fresh tensor's padding channels first
    const int ocs = (dst.w > 1 ? dst.sw : pad8(pl.OC)) >= pad8(pl.OC) && pl.OC % 8 ? pad8(pl.OC) : (pl.OC + 3) / 4 * 4;
    if (dst.w > 1 && dst.sw < ocs) return fail(DCV_EINVAL, "cl conv: destination pixel pitch %lld < %d stored channels", (long long)dst.sw, ocs);
    if (thin && !accumulate && !stat && !gate) {
        // fused 3-D stem (cl_stem3d_kernel): Conv3d(<= 8 -> 32, 4x4x4, stride (1, 2, 2), padding (0, 1, 1)) forward on 64 x 64 frames
        static const bool no_stem = getenv("DCV_CL_NO_STEM3") != nullptr;      // A/B only
        if (!no_stem && which == 0 && !g->transposed && g->kd == 4 && g->kh == 4 && g->kw == 4 && g->sd == 1 && g->sh == 2 && g->sw == 2 && g->pd == 0 && g->ph == 1 && g->pw == 1 &&
            src.h == 64 && src.w == 64 && dst.h == 32 && dst.w == 32 && dst.d == src.d - 3 && pl.OC == 32 && ocs == 32 &&
            (reinterpret_cast<uintptr_t>(src_p) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst_p) & 15) == 0 && src.sw >= 8 && dst.sw >= 32 &&
            cl_extent_bytes(src, 8) < (1ll << 31) && cl_extent_bytes(dst, 32) < (1ll << 31)) {
            ClStem3Args t;
            memset(&t, 0, sizeof(t));
            t.x = static_cast<const cl_h*>(src_p); t.y = static_cast<cl_h*>(dst_p); t.wp = static_cast<const cl_h*>(packed);
            t.N = src.n; t.OD = dst.d; t.act = act; t.slope = slope;
            t.x_sn = src.sn; t.y_sn = dst.sn; t.x_sd = (int32_t)src.sd; t.x_sh = (int32_t)src.sh; t.x_sw = (int32_t)src.sw;
            t.y_sd = (int32_t)dst.sd; t.y_sh = (int32_t)dst.sh; t.y_sw = (int32_t)dst.sw;
            t.x_bytes = (uint32_t)cl_extent_bytes(src, 8); t.y_bytes = (uint32_t)cl_extent_bytes(dst, 32);
            const unsigned nwg = (unsigned)(src.n * 8 * dst.d);      // (sample, band) pairs are a multiple of 8: every XCD slot sequence is whole
            hipLaunchKernelGGL(cl_stem3d_kernel, dim3(nwg), dim3(256), 0, static_cast<hipStream_t>(stream), t);
            snprintf(g_last_kernel, sizeof(g_last_kernel), "cl_stem3d_kernel (fused, thin source, " CL_HALF_NAME " channels-last)");
            DCV_LAUNCH_CHECK();
            return DCV_OK;
        }
    }
    if (thin && !accumulate && !stat) {
        // fused form (cl_widen3x3_kernel): 2-D 3x3 / stride 1 / pad 1 direct form out of a thin source on 64-wide rows, 64 or 128 destination channels
        static const bool no_fused = getenv("DCV_CL_NO_WIDEN3") != nullptr;      // A/B only
        const bool direct3 = (which == 0 && !g->transposed) || (which == 1 && g->transposed);
        if (!no_fused && direct3 && g->kd == 1 && g->kh == 3 && g->kw == 3 && g->sd == 1 && g->sh == 1 && g->sw == 1 && g->pd == 0 && g->ph == 1 && g->pw == 1 &&
            src.d == 1 && dst.d == 1 && src.w == 64 && src.h % 16 == 0 && dst.w == 64 && dst.h == src.h && (pl.OC == 64 || pl.OC == 128) && pl.OC == ocs &&
            (reinterpret_cast<uintptr_t>(src_p) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst_p) & 15) == 0 && src.sw >= 8 && dst.sw >= pl.OC &&
            cl_extent_bytes(src, 8) < (1ll << 31) && cl_extent_bytes(dst, pl.OC) < (1ll << 31)) {
            ClWiden3Args t;
            memset(&t, 0, sizeof(t));
            t.x = static_cast<const cl_h*>(src_p); t.y = static_cast<cl_h*>(dst_p); t.wp = static_cast<const cl_h*>(packed);
            t.N = src.n; t.H = src.h; t.OCp = (pl.OC + cl_pick_tile(pl.OC).bn - 1) / cl_pick_tile(pl.OC).bn * cl_pick_tile(pl.OC).bn; t.act = act; t.bands = src.h / 16; t.slope = slope;
            t.x_sn = src.sn; t.y_sn = dst.sn; t.x_sh = (int32_t)src.sh; t.x_sw = (int32_t)src.sw; t.y_sh = (int32_t)dst.sh; t.y_sw = (int32_t)dst.sw;
            t.x_bytes = (uint32_t)cl_extent_bytes(src, 8); t.y_bytes = (uint32_t)cl_extent_bytes(dst, pl.OC);
            const unsigned nwg = (unsigned)((src.n + 7) / 8 * 8 * t.bands);
            hipStream_t st3 = static_cast<hipStream_t>(stream);
            if (pl.OC == 128) hipLaunchKernelGGL((cl_widen3x3_kernel<4>), dim3(nwg), dim3(256), 0, st3, t);
            else hipLaunchKernelGGL((cl_widen3x3_kernel<2>), dim3(nwg), dim3(256), 0, st3, t);
            snprintf(g_last_kernel, sizeof(g_last_kernel), "cl_widen3x3_kernel<%d> (fused, thin source, " CL_HALF_NAME " channels-last)", pl.OC / 32);
            DCV_LAUNCH_CHECK();
            return DCV_OK;
        }
    }
    if (!thin && !accumulate && !gate) {
        // patch-staged scatter form (cl_patch_convt_kernel): all four stride-parity classes of a 4x4 / stride-2 layer from one staged source patch
        const ClPatchPlan pp = cl_patch_plan(which, g, pl, src, dst);
        const int64_t xb2 = cl_extent_bytes(src, Cp), yb2 = cl_extent_bytes(dst, ocs);
        if (pp.ok && ocs == pl.OC && (reinterpret_cast<uintptr_t>(src_p) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst_p) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(packed) & 15) == 0 && src.sw >= Cp && dst.sw >= pl.OC && xb2 < (1ll << 31) && yb2 < (1ll << 31) && pp.pack_bytes < (1ull << 31)) {
            ClPatchArgs t;
            memset(&t, 0, sizeof(t));
            t.x = static_cast<const cl_h*>(src_p); t.y = static_cast<cl_h*>(dst_p);
            t.wp = reinterpret_cast<const cl_h*>(static_cast<const char*>(packed) + cl_generic_pack_bytes(pl));
            t.N = src.n; t.H = src.h; t.W = src.w; t.C32 = pp.C32; t.OCp = pp.OCp; t.octiles = pp.octiles; t.NI = pp.NI; t.PRI = pp.PRI;
            t.RI = pp.RI; t.prows = pp.prows; t.bands = pp.bands; t.npatch = pp.npatch; t.wlog = pp.wlog; t.plog = pp.plog; t.act = act; t.y_c = pl.OC;
            t.slope = slope; t.total = pp.npatch * pp.octiles;
            t.x_sn = src.sn; t.y_sn = dst.sn; t.x_sh = (int32_t)src.sh; t.x_sw = (int32_t)src.sw; t.y_sh = (int32_t)dst.sh; t.y_sw = (int32_t)dst.sw;
            t.x_bytes = (uint32_t)xb2; t.y_bytes = (uint32_t)yb2; t.w_bytes = (uint32_t)pp.pack_bytes;
            if (stat && which == 0 && act == DCV_ACT_NONE && (size_t)pp.npatch * pp.OCp * 2 * sizeof(float) <= stat_bytes) {
                t.stat = stat;
                if (nparts) *nparts = pp.npatch;
                if (pitch) *pitch = pp.OCp;
            }
            if (pp.NP == 256) hipLaunchKernelGGL((cl_patch_convt_kernel<8>), dim3((unsigned)((t.total + 7) / 8 * 8)), dim3(512), 0, static_cast<hipStream_t>(stream), t);
            else hipLaunchKernelGGL((cl_patch_convt_kernel<4>), dim3((unsigned)((t.total + 7) / 8 * 8)), dim3(256), 0, static_cast<hipStream_t>(stream), t);
            snprintf(g_last_kernel, sizeof(g_last_kernel), "cl_patch_convt_kernel<%d waves> (4 classes from one staged patch, %d x %d source, " CL_HALF_NAME " channels-last)", pp.NP / 32, src.h, src.w);
            DCV_LAUNCH_CHECK();
            return DCV_OK;
        }
    }
    const ClTile tc = cl_pick_tile(pl.OC);
    const int OCp = (pl.OC + tc.bn - 1) / tc.bn * tc.bn;
    const int64_t xb = cl_extent_bytes(src, thin ? 8 : Cp), yb = cl_extent_bytes(dst, ocs);
    if (xb >= (1ll << 31) || yb >= (1ll << 31)) return fail(DCV_EUNSUPPORTED, "cl conv: tensors beyond 2 GB need per-sample descriptors");
    ClGatherPack pk;
    memset(&pk, 0, sizeof(pk));
    size_t off = 0;
    int n = 0;
    int64_t maxtm = 0;
    for (const ClClass& c : pl.cls) {
        const int T = c.t[0].n * c.t[1].n * c.t[2].n;
        const size_t b = cl_class_pack_bytes(c, pl.RC, pl.OC);
        if (T == 0 || c.o_ext[0] <= 0 || c.o_ext[1] <= 0 || c.o_ext[2] <= 0) {
            if (T != 0) off += b;
            if (T == 0 && c.o_ext[0] > 0 && c.o_ext[1] > 0 && c.o_ext[2] > 0 && !accumulate) return fail(DCV_EUNSUPPORTED, "cl conv: a position class without taps");
            continue;
        }
        if (T > 64) return fail(DCV_EUNSUPPORTED, "cl conv: more than 64 taps");
        ClGatherArgs& a = pk.c[n];
        a.x = static_cast<const cl_h*>(src_p);
        a.y = static_cast<cl_h*>(dst_p);
        a.wp = reinterpret_cast<const cl_h*>(static_cast<const char*>(packed) + off);
        off += b;
        const int64_t M64 = (int64_t)dst.n * c.o_ext[0] * c.o_ext[1] * c.o_ext[2];
        if (M64 >= (1ll << 31)) return fail(DCV_EUNSUPPORTED, "cl conv: too many positions");
        a.M = (int)M64; a.OCp = OCp; a.T = T;
        a.cblk = thin ? 0 : Cp / 32;
        a.nsteps = thin ? (T + 3) / 4 : T * (Cp / 32);
        a.y_c = ocs;
        a.x_cmax = 2 * pad8(pl.RC);
        a.div_sp = make_fastdiv((uint32_t)(c.o_ext[0] * c.o_ext[1] * c.o_ext[2]));
        a.div_hw = make_fastdiv((uint32_t)(c.o_ext[1] * c.o_ext[2]));
        a.div_w = make_fastdiv((uint32_t)c.o_ext[2]);
        a.td = c.t[0]; a.th = c.t[1]; a.tw = c.t[2];
        a.x_sn = src.sn; a.x_sd = (int32_t)src.sd; a.x_sh = (int32_t)src.sh; a.x_sw = (int32_t)src.sw;
        a.y_sn = dst.sn;
        a.y_sd = (int32_t)(dst.sd * c.out_mul[0]); a.y_sh = (int32_t)(dst.sh * c.out_mul[1]); a.y_sw = (int32_t)(dst.sw * c.out_mul[2]);
        a.y_off = (int32_t)(dst.sd * c.out_off[0] + dst.sh * c.out_off[1] + dst.sw * c.out_off[2]);
        a.act = act; a.slope = slope; a.accumulate = accumulate;
        // 16-byte row-order stores need whole 8-channel groups at 16-byte-aligned pixel bases (pitches are multiples of 8 elements: cl_check_tensor)
        static const bool no_coalesce = getenv("DCV_CL_DIRECT_EPILOGUE") != nullptr;      // A/B only
        a.coalesce = (ocs % 8 == 0 && (reinterpret_cast<uintptr_t>(dst_p) & 15) == 0 && !no_coalesce) ? 1 : 0;
        if (gate) {
            if (!a.coalesce) return fail(DCV_EUNSUPPORTED, "cl conv: the gated epilogue needs the row-order store form");
            a.gate = static_cast<const cl_h*>(gate); a.gate_slope = gate_slope;
        }
        a.x_bytes = (uint32_t)xb; a.y_bytes = (uint32_t)yb;
        for (int t = 0; t < T; ++t) {
            const int ud = t / (c.t[1].n * c.t[2].n), uh = (t / c.t[2].n) % c.t[1].n, uw = t % c.t[2].n;
            a.toff[t] = (int32_t)(2 * ((int64_t)c.t[0].delta[ud] * src.sd + (int64_t)c.t[1].delta[uh] * src.sh + (int64_t)c.t[2].delta[uw] * src.sw));
            a.tsel[t] = ud | (uh << 2) | (uw << 4);
        }
        maxtm = std::max<int64_t>(maxtm, (M64 + tc.bm - 1) / tc.bm);
        if (++n > 4) return fail(DCV_EUNSUPPORTED, "cl conv: more than 4 position classes");
    }
    if (n == 0) return DCV_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (stat && which == 0 && act == DCV_ACT_NONE && !accumulate) {
        const size_t need = (size_t)n * maxtm * OCp * 2 * sizeof(float);
        if (need <= stat_bytes) {
            // rows of position tiles a class does not have (classes of different sizes) must read as zero
            DCV_HIP_CHECK(hipMemsetAsync(stat, 0, need, st));
            for (int i = 0; i < n; ++i) { pk.c[i].stat = stat; pk.c[i].stat_row0 = (int32_t)(i * maxtm); }
            if (nparts) *nparts = (int)(n * maxtm);
            if (pitch) *pitch = OCp;
        }
    }
#ifdef DCV_DEBUG_TIMING      // timing-experiment builds only: the shipped library cannot be told to drop its stores
    static const int dbg = getenv("DCV_CL_DEBUG") ? atoi(getenv("DCV_CL_DEBUG")) : 0;      // 1: no epilogue stores
#else
    constexpr int dbg = 0;
#endif
    for (int i = 0; i < n; ++i) pk.c[i].pad2 = dbg;
    ClSplitK sk = {0, 0, 0, 0};
    if (!stat && !accumulate && !gate) sk = cl_splitk_plan(n, thin, pk.c[0].M, (int)pk.c[0].div_sp.div, OCp, pk.c[0].nsteps, tc, ocs);
    if (sk.KS && (!ws || ws_bytes < sk.slab_bytes || (reinterpret_cast<uintptr_t>(ws) & 15))) sk.KS = 0;
    if (sk.KS) { pk.c[0].slab = static_cast<float*>(ws); pk.c[0].ks_per = sk.ks_per; pk.c[0].slab_m = sk.slab_m; }
    for (int i = n; i < 4; ++i) pk.c[i] = pk.c[0];
    pk.ncls = n; pk.tiles_oc = OCp / tc.bn; pk.tiles_m = (int)maxtm;
    dim3 grid((unsigned)((maxtm + 7) / 8 * 8 * pk.tiles_oc * n));
    if (sk.KS) grid.y = (unsigned)sk.KS;
    cl_launch_tile(tc, pk, thin, grid, st);
    if (sk.KS) {
        DCV_LAUNCH_CHECK();
        const int64_t tot = (int64_t)pk.c[0].M * (ocs / 8);
        hipLaunchKernelGGL(cl_splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, pk.c[0], sk.KS);
        snprintf(g_last_kernel, sizeof(g_last_kernel), "cl_gather_kernel<%d x %d tile> (1 class, split-K x %d, " CL_HALF_NAME " channels-last)", tc.bn, tc.bm, sk.KS);
        DCV_LAUNCH_CHECK();
        return DCV_OK;
    }
    snprintf(g_last_kernel, sizeof(g_last_kernel), "cl_gather_kernel<%d x %d tile%s> (%d class%s, " CL_HALF_NAME " channels-last)", tc.bn, tc.bm, thin ? ", thin" : "", n, n == 1 ? "" : "es");
    DCV_LAUNCH_CHECK();
    return DCV_OK;
}

int dcv_cl_conv_forward(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* packed, void* y, const dcv_dims5* yd,
                        int act, float slope, void* ws, size_t ws_bytes, void* stream) {
    return cl_conv_run(0, g, x, xd, packed, y, yd, act, slope, 0, ws, ws_bytes, stream);
}
// bytes of the per-tile BatchNorm sums dcv_cl_conv_forward_stats leaves (0: this geometry's form does not produce them)
size_t dcv_cl_conv_stats_bytes(const dcv_conv_geom* g, const dcv_dims5* xd, const dcv_dims5* yd) {
    if (!g || !xd || !yd) return 0;
    ClPlan pl;
    if (cl_make_plan(0, g, xd, yd, &pl) != DCV_OK || cl_thin_out(pl, g)) return 0;
    const ClTile tc = cl_pick_tile(pl.OC);
    const int OCp = (pl.OC + tc.bn - 1) / tc.bn * tc.bn;
    int n = 0;
    int64_t maxtm = 0;
    for (const ClClass& c : pl.cls) {
        if (c.t[0].n * c.t[1].n * c.t[2].n == 0 || c.o_ext[0] <= 0 || c.o_ext[1] <= 0 || c.o_ext[2] <= 0) continue;
        const int64_t M64 = (int64_t)yd->n * c.o_ext[0] * c.o_ext[1] * c.o_ext[2];
        maxtm = std::max<int64_t>(maxtm, (M64 + tc.bm - 1) / tc.bm);
        ++n;
    }
    return (size_t)n * maxtm * OCp * 2 * sizeof(float);
}
int dcv_cl_conv_forward_stats(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* packed, void* y, const dcv_dims5* yd,
                              float* stat, size_t stat_bytes, int* nparts, int* pitch, void* ws, size_t ws_bytes, void* stream) {
    if (!stat || !nparts || !pitch) return fail(DCV_EINVAL, "cl_conv_forward_stats: null pointer");
    return cl_conv_run(0, g, x, xd, packed, y, yd, DCV_ACT_NONE, 0.f, 0, ws, ws_bytes, stream, stat, stat_bytes, nparts, pitch);
}
int dcv_cl_conv_backward_data(const dcv_conv_geom* g, const void* dy, const dcv_dims5* dyd, const void* packed, void* dx, const dcv_dims5* dxd,
                              int accumulate, void* ws, size_t ws_bytes, void* stream) {
    return cl_conv_run(1, g, dy, dxd, packed, dx, dyd, DCV_ACT_NONE, 0.f, accumulate, ws, ws_bytes, stream);
}

// backward_data followed by the (Leaky)ReLU derivative of the layer that produced this convolution's input, read off that input (`xg`: dx's shape AND strides):
// dx = (accumulate ? dx : 0) + conv^T(dy, w); dx *= (xg > 0 ? 1 : slope).  Replaces the separate derivative pass of a conv + LeakyReLU pair whose output feeds this
// convolution (Inconv -> DownBlock 0, generator.py:173-176,203-207), as dcv_conv_backward_data_gated does on the fp32 path.  DCV_EUNSUPPORTED before any launch
// where the form has no such epilogue (thin operands, unaligned destinations): the caller then runs the two steps separately.
int dcv_cl_conv_backward_data_gated(const dcv_conv_geom* g, const void* dy, const dcv_dims5* dyd, const void* packed, void* dx, const dcv_dims5* dxd,
                                    int accumulate, const void* xg, const dcv_dims5* xgd, int act, float slope, void* ws, size_t ws_bytes, void* stream) {
    if (!xg || !xgd || !dxd) return fail(DCV_EINVAL, "cl_conv_backward_data_gated: null pointer");
    if (act != DCV_ACT_LEAKY) return fail(DCV_EUNSUPPORTED, "cl_conv_backward_data_gated: (Leaky)ReLU only");
    if (!same_shape(*xgd, *dxd) || xgd->sn != dxd->sn || xgd->sc != dxd->sc || xgd->sd != dxd->sd || xgd->sh != dxd->sh || xgd->sw != dxd->sw)
        return fail(DCV_EINVAL, "cl_conv_backward_data_gated: the gate tensor must have dx's shape and strides");
    return cl_conv_run(1, g, dy, dxd, packed, dx, dyd, DCV_ACT_NONE, 0.f, accumulate, ws, ws_bytes, stream, nullptr, 0, nullptr, nullptr, xg, slope);
}

size_t dcv_cl_wgrad_workspace_bytes(const dcv_conv_geom* g, const dcv_dims5* x, const dcv_dims5* y) {
    if (!g || !x || !y) return 0;
    ClWgradPlan p;
    if (cl_wgrad_plan(g, g->transposed ? *x : *y, g->transposed ? *y : *x, &p) != DCV_OK) return 0;
    return p.tab_bytes + p.slab_bytes + 512;
}

// dw (fp32, torch layout) = corr(x, dy); x and dy bf16 channels-last.  conv: dense = dy, gathered = x; transposed conv: dense = x, gathered = dy.
static int cl_conv_backward_weight(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* dy, const dcv_dims5* dyd, float* dw, int accumulate,
                                   void* ws, size_t ws_bytes, void* stream) {
    if (!g || !x || !xd || !dy || !dyd || !dw || !ws) return fail(DCV_EINVAL, "cl_conv_backward_weight: null pointer");
    ClPlan chk;
    int rc = cl_make_plan(0, g, xd, dyd, &chk);      // validates the geometry
    if (rc != DCV_OK) return rc;
    const dcv_dims5& D = g->transposed ? *xd : *dyd;
    const dcv_dims5& G = g->transposed ? *dyd : *xd;
    const cl_h* dp = static_cast<const cl_h*>(g->transposed ? x : dy);
    const cl_h* gp = static_cast<const cl_h*>(g->transposed ? dy : x);
    if ((rc = cl_check_tensor(D, "cl wgrad dense operand")) != DCV_OK || (rc = cl_check_tensor(G, "cl wgrad gathered operand")) != DCV_OK) return rc;
    int64_t dpitch = 0;
    if (!cl_pixel_linear(D, &dpitch)) return fail(DCV_EUNSUPPORTED, "cl wgrad: the dense operand must be pixel-linear (a whole tensor or a channel slice of one)");
    ClWgradPlan p;
    if ((rc = cl_wgrad_plan(g, D, G, &p)) != DCV_OK) return rc;
    if (ws_bytes < p.tab_bytes + p.slab_bytes) return fail(DCV_EWORKSPACE, "cl wgrad: workspace too small (%zu needed, %zu given)", p.tab_bytes + p.slab_bytes, ws_bytes);
    const int64_t dbytes = 2 * ((p.M - 1) * dpitch + pad8(D.c)), gbytes = cl_extent_bytes(G, pad8(G.c));
    if (dbytes >= (1ll << 31) || gbytes >= (1ll << 31)) return fail(DCV_EUNSUPPORTED, "cl wgrad: tensors beyond 2 GB");
    if (G.w > 1 && G.sw < pad8(G.c)) return fail(DCV_EINVAL, "cl wgrad: gathered operand's pixel pitch below its padded channel count");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ClPosEntry* tab = static_cast<ClPosEntry*>(ws);
    float* slab = reinterpret_cast<float*>(static_cast<char*>(ws) + p.tab_bytes);
    // A kept table is published (inserted into g_tabs) only AFTER its build kernel has been launched and has completed: a second host thread with the same
    // key either finds a complete table or builds its own (the loser of the insert frees its copy); a failed launch / synchronise frees the copy and leaves
    // no entry behind.  The cache is bounded (CL_TAB_CACHE_MAX keys: the three GPU configs hold ~30 each); past the bound a call builds into its workspace.
    bool build = true, keep = false;
    static const bool no_cache = getenv("DCV_CL_NO_POSTAB_CACHE") != nullptr;
    ClTabKey key;
    memset(&key, 0, sizeof(key));
    if (!no_cache) {
        DCV_HIP_CHECK(hipGetDevice(&key.dev));
        const int32_t v[24] = {g->kd, g->kh, g->kw, g->sd, g->sh, g->sw, g->pd, g->ph, g->pw, D.n, D.d, D.h, D.w, G.d, G.h, G.w, (int32_t)G.sd, (int32_t)G.sh, (int32_t)G.sw, 0, 0, 0, 0, 0};
        memcpy(key.v, v, sizeof(v));
        key.w[0] = G.sn;
        bool room = false;
        {
            std::lock_guard<std::mutex> lk(g_tab_mu);
            auto it = g_tabs.find(key);
            if (it != g_tabs.end()) { tab = it->second; build = false; }
            else room = g_tabs.size() < CL_TAB_CACHE_MAX;
        }
        if (build && room) {
            ClPosEntry* dev_tab = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&dev_tab), p.tab_bytes) == hipSuccess) { tab = dev_tab; keep = true; }
            else (void)hipGetLastError();      // no memory for a kept copy: this call builds into the workspace
        }
    }
    if (build) {
        ClPosArgs a;
        memset(&a, 0, sizeof(a));
        a.tab = tab; a.M = (int)p.M; a.T = p.T; a.KH = g->kh; a.KW = g->kw;
        a.div_sp = make_fastdiv((uint32_t)(D.d * D.h * D.w)); a.div_hw = make_fastdiv((uint32_t)(D.h * D.w)); a.div_w = make_fastdiv((uint32_t)D.w);
        a.s[0] = g->sd; a.s[1] = g->sh; a.s[2] = g->sw; a.p[0] = g->pd; a.p[1] = g->ph; a.p[2] = g->pw; a.k[0] = g->kd; a.k[1] = g->kh; a.k[2] = g->kw;
        a.gext[0] = G.d; a.gext[1] = G.h; a.gext[2] = G.w;
        a.g_sn = G.sn; a.g_sd = (int32_t)G.sd; a.g_sh = (int32_t)G.sh; a.g_sw = (int32_t)G.sw;
        hipLaunchKernelGGL(cl_postab_kernel, dim3((unsigned)((p.M + 255) / 256)), dim3(256), 0, st, a);
        g_launches.fetch_add(1, std::memory_order_relaxed);
        hipError_t e = hipGetLastError();
        // a kept table is read by later calls on OTHER streams (the discriminators' lanes): it must be complete before it is published (once per layer)
        if (e == hipSuccess && keep) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            if (keep) (void)hipFree(tab);
            return fail(DCV_EHIP, "cl wgrad: building the position table failed: %s", hipGetErrorString(e));
        }
        if (keep) {
            std::lock_guard<std::mutex> lk(g_tab_mu);
            auto ins = g_tabs.emplace(key, tab);
            if (!ins.second) { (void)hipFree(tab); tab = ins.first->second; }      // another thread published the same table meanwhile: use that one
        }
    }
    {
        ClWgradArgs a;
        memset(&a, 0, sizeof(a));
        a.d = dp; a.g = gp; a.tab = tab; a.slab = slab;
        a.M = (int)p.M; a.chunk = p.chunk;
        a.d_pitch2 = (int32_t)(2 * dpitch); a.d_cbytes = 2 * pad8(D.c); a.g_cbytes = 2 * pad8(G.c); a.T = p.T;
        a.tiles_d = p.tiles_d; a.gblocks = p.gblocks; a.gcb8 = p.gcb / 8; a.ntpt = p.ntpt;
        a.d_bytes = (uint32_t)dbytes; a.g_bytes = (uint32_t)gbytes;
        for (int t = 0; t < p.T; ++t) {
            const int kd = t / (g->kh * g->kw), kh = (t / g->kw) % g->kh, kw = t % g->kw;
            a.toff[t] = (int32_t)(2 * ((int64_t)kd * G.sd + (int64_t)kh * G.sh + (int64_t)kw * G.sw));
        }
        static const bool flat = getenv("DCV_CL_WGRAD_FLAT") != nullptr;      // A/B only
        a.tiles = p.tiles; a.S = p.S; a.xcd_map = flat ? 0 : 1; a.wtiles = p.wtiles;
        if (p.narrow) hipLaunchKernelGGL((cl_wgrad_kernel<true>), dim3((unsigned)(p.wtiles * p.S)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((cl_wgrad_kernel<false>), dim3((unsigned)(p.wtiles * p.S)), dim3(256), 0, st, a);
        DCV_LAUNCH_CHECK();
    }
    {
        ClWgradReduceArgs a;
        memset(&a, 0, sizeof(a));
        a.slab = slab; a.dw = dw; a.S = p.S; a.tiles = p.tiles; a.tiles_d = p.tiles_d; a.gblocks = p.gblocks; a.gcb = p.gcb; a.ntpt = p.ntpt; a.T = p.T;
        a.DC = D.c; a.GC = G.c; a.ws_d = (int64_t)G.c * p.T; a.accumulate = accumulate ? 1 : 0;
        const int64_t tot = (int64_t)D.c * G.c * p.T;
        static const bool old_reduce = getenv("DCV_CL_OLD_WGRAD_REDUCE") != nullptr;      // A/B only
        const int64_t nel = (int64_t)p.tiles * (128 * 128);
        if (old_reduce) {
            a.lpe = (p.S >= 32 && tot * 64 < (1ll << 31)) ? 64 : 1;
            hipLaunchKernelGGL(cl_wgrad_reduce_kernel, dim3((unsigned)((tot * a.lpe + 255) / 256)), dim3(256), 0, st, a);
        } else if (p.S >= 64) {
            hipLaunchKernelGGL((cl_wgrad_reduce_slab_kernel<16, 16>), dim3((unsigned)((nel + 15) / 16)), dim3(256), 0, st, a);
        } else {
            hipLaunchKernelGGL((cl_wgrad_reduce_slab_kernel<64, 4>), dim3((unsigned)((nel + 63) / 64)), dim3(256), 0, st, a);
        }
        DCV_LAUNCH_CHECK();
    }
    snprintf(g_last_kernel, sizeof(g_last_kernel), "cl_wgrad_kernel (%d tiles%s x %d position splits, " CL_HALF_NAME " channels-last)", p.tiles, p.narrow ? " in pairs, 64 dense rows" : "", p.S);
    return DCV_OK;
}
int dcv_cl_conv_backward_weight(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* dy, const dcv_dims5* dyd, float* dw,
                                void* ws, size_t ws_bytes, void* stream) {
    return cl_conv_backward_weight(g, x, xd, dy, dyd, dw, 0, ws, ws_bytes, stream);
}
// dw = (accumulate ? dw : 0) + corr(x, dy): as dcv_conv_backward_weight_acc
int dcv_cl_conv_backward_weight_acc(const dcv_conv_geom* g, const void* x, const dcv_dims5* xd, const void* dy, const dcv_dims5* dyd, float* dw, int accumulate,
                                    void* ws, size_t ws_bytes, void* stream) {
    return cl_conv_backward_weight(g, x, xd, dy, dyd, dw, accumulate, ws, ws_bytes, stream);
}

}  // extern "C"
