// Diagnostic only (tools/lds_guard_probe.py): workgroups that hold an LDS pattern while other kernels run on the card, and count the words
// that changed under them.  A nonzero count means some other workgroup's write (an LDS-DMA that landed late or out of its range) reached LDS it does not own.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void lds_guard_kernel(unsigned long long* bad, uint32_t* first, int rounds, int sleep) {
    __shared__ uint32_t pat[8192];   // 32 KB
    const uint32_t salt = 0x9e3779b9u * (blockIdx.x + 1);
    for (int i = threadIdx.x; i < 8192; i += 256) pat[i] = salt ^ (uint32_t)i * 2654435761u;
    __syncthreads();
    unsigned long long n = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int s = 0; s < sleep; ++s) __builtin_amdgcn_s_sleep(127);
        for (int i = threadIdx.x; i < 8192; i += 256) {
            const uint32_t want = salt ^ (uint32_t)i * 2654435761u, got = pat[i];
            if (got != want) {
                if (n == 0 && atomicAdd((unsigned long long*)bad, 0ull) == 0) { first[0] = blockIdx.x; first[1] = (uint32_t)i; first[2] = got; first[3] = want; }
                ++n;
                pat[i] = want;
            }
        }
        __syncthreads();
    }
    if (n) atomicAdd(bad, n);
}

extern "C" int lds_guard_launch(void* bad, void* first, int blocks, int rounds, int sleep, void* stream) {
    hipLaunchKernelGGL(lds_guard_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)bad, (uint32_t*)first, rounds, sleep);
    return (int)hipGetLastError();
}

// Producer / consumer pair on ONE stream: fill writes pattern k, check reads it back in the next launch.  Words that do not read back as written, while
// other streams keep the card busy, mean a kernel saw memory its predecessor on the same stream had not (visibly) written yet.
__global__ __launch_bounds__(256) void mem_fill_kernel(uint32_t* buf, int64_t n, uint32_t k) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) buf[i] = (uint32_t)i * 2654435761u ^ (k * 0x9e3779b9u);
}
__global__ __launch_bounds__(256) void mem_check_kernel(const uint32_t* buf, int64_t n, uint32_t k, unsigned long long* bad, uint32_t* first) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t want = (uint32_t)i * 2654435761u ^ (k * 0x9e3779b9u), got = buf[i];
        if (got != want) {
            if (c == 0 && atomicAdd(bad, 0ull) == 0) { first[0] = (uint32_t)i; first[1] = got; first[2] = want; first[3] = k; }
            ++c;
        }
    }
    if (c) atomicAdd(bad, c);
}
extern "C" int mem_guard_round(void* buf, long long n, unsigned k, void* bad, void* first, int blocks_fill, int blocks_check, void* stream) {
    hipLaunchKernelGGL(mem_fill_kernel, dim3(blocks_fill), dim3(256), 0, (hipStream_t)stream, (uint32_t*)buf, (int64_t)n, k);
    hipLaunchKernelGGL(mem_check_kernel, dim3(blocks_check), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)buf, (int64_t)n, k, (unsigned long long*)bad, (uint32_t*)first);
    return (int)hipGetLastError();
}
extern "C" int mem_guard_check(void* buf, long long n, unsigned k, void* bad, void* first, int blocks_check, void* stream) {
    hipLaunchKernelGGL(mem_check_kernel, dim3(blocks_check), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)buf, (int64_t)n, k, (unsigned long long*)bad, (uint32_t*)first);
    return (int)hipGetLastError();
}

// The cross-wave hand-off of the small kernels (thin_rows_kernel, block_sum): waves 1-3 leave values in LDS, one barrier, wave 0 reads them.
__global__ __launch_bounds__(256) void xwave_guard_kernel(unsigned long long* bad, uint32_t* first, uint32_t salt) {
    __shared__ uint32_t red[3][12][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave > 0)
        for (int i = 0; i < 12; ++i) red[wave - 1][i][lane] = (salt + blockIdx.x) * 2654435761u ^ (uint32_t)((wave * 12 + i) * 64 + lane);
    __syncthreads();
    if (wave != 0) return;
    unsigned long long n = 0;
    for (int w = 1; w < 4; ++w)
        for (int i = 0; i < 12; ++i) {
            const uint32_t want = (salt + blockIdx.x) * 2654435761u ^ (uint32_t)((w * 12 + i) * 64 + lane), got = red[w - 1][i][lane];
            if (got != want) {
                if (n == 0 && atomicAdd(bad, 0ull) == 0) { first[0] = blockIdx.x; first[1] = (uint32_t)((w * 12 + i) * 64 + lane); first[2] = got; first[3] = want; }
                ++n;
            }
        }
    if (n) atomicAdd(bad, n);
}
extern "C" int xwave_guard_launch(void* bad, void* first, int blocks, unsigned salt, void* stream) {
    hipLaunchKernelGGL(xwave_guard_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)bad, (uint32_t*)first, salt);
    return (int)hipGetLastError();
}

// The same producer / consumer pair with DIFFERENT block -> data maps: block b fills 4 KiB chunk b, b + G, ..; the checker's block b reads chunk b + shift, so a
// line is written by a workgroup of one XCD (ids go round-robin over the 8 XCDs) and read in the next launch by a workgroup of another.
__global__ __launch_bounds__(256) void mem_fill_chunks_kernel(uint32_t* buf, int64_t nchunks, uint32_t k) {
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x)
        for (int i = threadIdx.x; i < 1024; i += 256) { const int64_t j = c * 1024 + i; buf[j] = (uint32_t)j * 2654435761u ^ (k * 0x9e3779b9u); }
}
__global__ __launch_bounds__(256) void mem_check_chunks_kernel(const uint32_t* buf, int64_t nchunks, uint32_t k, int shift, unsigned long long* bad, uint32_t* first) {
    unsigned long long n = 0;
    for (int64_t c0 = blockIdx.x; c0 < nchunks; c0 += gridDim.x) {
        const int64_t c = (c0 + shift) % nchunks;
        for (int i = threadIdx.x; i < 1024; i += 256) {
            const int64_t j = c * 1024 + i;
            const uint32_t want = (uint32_t)j * 2654435761u ^ (k * 0x9e3779b9u), got = buf[j];
            if (got != want) {
                if (n == 0 && atomicAdd(bad, 0ull) == 0) { first[0] = (uint32_t)j; first[1] = got; first[2] = want; first[3] = k; }
                ++n;
            }
        }
    }
    if (n) atomicAdd(bad, n);
}
extern "C" int mem_guard_round_xcd(void* buf, long long nchunks, unsigned k, int shift, void* bad, void* first, int blocks, void* stream) {
    hipLaunchKernelGGL(mem_fill_chunks_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t*)buf, (int64_t)nchunks, k);
    hipLaunchKernelGGL(mem_check_chunks_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)buf, (int64_t)nchunks, k, shift, (unsigned long long*)bad, (uint32_t*)first);
    return (int)hipGetLastError();
}
