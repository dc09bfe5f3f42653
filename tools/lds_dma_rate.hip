// What one CU's vector-memory path delivers into LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave instruction) when the lines are already in the L2 (or in the
// CU's own L1): the ceiling of every gather kernel's staging, measured bare and beside the MFMAs of a K step.  Stand-alone: hipcc --offload-arch=gfx950 -O3 -o lds_dma_rate.bin
//   region = bytes each workgroup walks (8 KB: the CU's L1 holds it; 1 MB shared by all workgroups: L2 hits); W = waves per workgroup; wgs per CU; MFMA = 32x32x16 bf16 per 1 KB piece.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;

// PAT 0: a piece = 1 KB contiguous (8 whole 128-byte lines).  PAT 1: a piece = 16 rows x 64 bytes at a 256-byte row pitch (HALF of 16 lines: a 32-channel K step of a
// channels-last gather); the other halves are the next iteration's pieces.  PAT 2: 8 rows x 128 bytes at the same pitch (whole lines: a 64-channel K step).
template <int W, int MFMA_PER_PIECE, int PAT = 0>
__global__ __launch_bounds__(64 * W) void k(const char* src, uint32_t src_bytes, uint32_t region, int shared_region, float* out, int iters) {
    extern __shared__ char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, src_bytes, 0x00020000);
    const uint32_t base = shared_region ? 0u : (blockIdx.x * region) % (src_bytes - region);
    f32x16 acc[2] = {{0}, {0}};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.001f * lane + i); fb[i] = (__bf16)(1.0f - 0.002f * lane); }
    uint32_t off = wave * 4096u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t o, vo;
            if constexpr (PAT == 0) { o = base + ((off + u * 1024u) & (region - 1)); vo = lane * 16u; }
            else if constexpr (PAT == 1) { o = base + ((off * 4 + u * 4096u) & (region - 1)) + (it & 1) * 64u; vo = (lane >> 2) * 256u + (lane & 3) * 16u; }   // 16 rows of this piece, half (it & 1)
            else { o = base + ((off * 2 + u * 2048u) & (region - 1)); vo = (lane >> 3) * 256u + (lane & 7) * 16u; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + ((it & 1) * W * 4 + wave * 4 + u) * 1024), 16, vo, o, 0, 0);
#pragma unroll
            for (int m = 0; m < MFMA_PER_PIECE; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 1], 0, 0, 0);
        }
        if (PAT != 1 || (it & 1)) off += W * 4096u;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // the previous round's four pieces have landed; this round's stay in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    out[blockIdx.x * 64 * W + threadIdx.x] = s + smem[threadIdx.x * 4];
}

template <int W, int M, int PAT = 0>
static void run(const char* src, uint32_t src_bytes, uint32_t region, int shared_region, int wgs_per_cu, float* out) {
    const int iters = 4000, grid = 256 * wgs_per_cu;
    const size_t lds = (size_t)2 * W * 4 * 1024;
    hipFuncSetAttribute((const void*)k<W, M, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<W, M, PAT>), dim3(grid), dim3(64 * W), lds, 0, src, src_bytes, region, shared_region, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<W, M, PAT>), dim3(grid), dim3(64 * W), lds, 0, src, src_bytes, region, shared_region, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * W * iters * 4096.0, flops = (double)grid * W * iters * 4.0 * M * 32 * 32 * 16 * 2;
    printf("pattern %d  waves/wg %d  wgs/CU %d  region %7u B %-6s  mfma/piece %d : %7.2f TB/s chip = %6.1f GB/s per CU = %5.1f B/clk/CU at 2.4 GHz ; %7.1f TFLOP/s ; %.3f ms\n", PAT, W, wgs_per_cu, region,
           shared_region ? "shared" : "own", M, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256, bytes / ms * 1e-6 / 256 / 2.4, flops / ms * 1e-9, ms);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); exit(1); }
}

int main() {
    const uint32_t src_bytes = 64u << 20;
    char* src; float* out;
    if (hipMalloc(&src, src_bytes) != hipSuccess || hipMalloc(&out, 256 * 8 * 512 * 4) != hipSuccess) return 1;
    hipMemset(src, 1, src_bytes);
    for (int sh = 0; sh < 2; ++sh) {
        const uint32_t region = sh ? (1u << 20) : (8u << 10);
        run<4, 0>(src, src_bytes, region, sh, 1, out);
        run<4, 0>(src, src_bytes, region, sh, 2, out);
        run<4, 0>(src, src_bytes, region, sh, 4, out);
        run<8, 0>(src, src_bytes, region, sh, 1, out);
        run<8, 0>(src, src_bytes, region, sh, 2, out);
        run<4, 1>(src, src_bytes, region, sh, 2, out);
        run<4, 2>(src, src_bytes, region, sh, 2, out);
        run<4, 4>(src, src_bytes, region, sh, 2, out);
        run<8, 2>(src, src_bytes, region, sh, 1, out);
        run<8, 4>(src, src_bytes, region, sh, 1, out);
    }
    for (int sh = 0; sh < 2; ++sh) {      // channels-last gather patterns: half lines (32-channel K steps) against whole lines (64-channel K steps)
        const uint32_t region = sh ? (4u << 20) : (64u << 10);
        run<4, 0, 1>(src, src_bytes, region, sh, 2, out);
        run<4, 0, 2>(src, src_bytes, region, sh, 2, out);
        run<4, 2, 1>(src, src_bytes, region, sh, 2, out);
        run<4, 2, 2>(src, src_bytes, region, sh, 2, out);
        run<4, 4, 2>(src, src_bytes, region, sh, 2, out);
    }
    run<4, 0>(src, src_bytes, 32u << 20, 1, 2, out);      // L2 misses on most XCDs: MALL / HBM
    run<4, 2>(src, src_bytes, 32u << 20, 1, 2, out);
    return 0;
}
