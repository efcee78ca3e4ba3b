// ubench_gather.hip -- calibrates rocprofv3's FETCH_SIZE for the access pattern of k_accumulate: every lane reads one
// random 64-byte row (4 x global_load_dwordx4) of a table far larger than the Infinity Cache.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_gather tools/ubench_gather.hip
// run:   rocprofv3 --pmc FETCH_SIZE -d gpurun_out/calib -o g --output-format csv -- ./gpurun_out/ubench_gather
// The kernel requests exactly rows_read * 64 bytes; FETCH_SIZE (KB) * 1024 / that = the factor to apply.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ table, uint64_t rows, unsigned iters, uint32_t *__restrict__ out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = t * 0x9E3779B97F4A7C15ull + 12345;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (unsigned i = 0; i < iters; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t row = (s >> 20) % rows;
        const uint4 *p = table + row * 4;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
        acc.y += a.y + b.z + c.w + d.x;
    }
    out[t] = acc.x ^ acc.y;
}

__global__ void __launch_bounds__(256) k_stream(const uint4 *__restrict__ table, uint64_t count, uint32_t *__restrict__ out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint64_t i = t; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 a = table[i];
        acc.x ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    out[t] = acc.x;
}

// the read pattern of k_ntt_pass (ntt.hip): a 1024-element tile = 4 adjacent sub-transforms x 256 inputs at stride S = n / 256 elements of
// 32 bytes; lane e reads element (blk0 + e % 4) + (e / 4) * S as two global_load_dwordx4, so four lanes cover one 128-byte piece
__global__ void __launch_bounds__(512) k_pieces(const uint4 *__restrict__ x, unsigned log_n, uint32_t *__restrict__ out)
{
    const unsigned S = (1u << log_n) >> 8, blk0 = blockIdx.x * 4;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (unsigned e = threadIdx.x; e < 1024; e += 512) {
        const unsigned b = e % 4, i = e / 4;
        const uint4 *p = x + ((size_t)(blk0 + b) + (size_t)i * S) * 2;
        uint4 lo = p[0], hi = p[1];
        acc.x ^= lo.x ^ hi.y;
        acc.y += lo.z + hi.w;
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc.x ^ acc.y;
}

int main()
{
    const uint64_t rows = (uint64_t)3 << 26; // 2^26 * 3 rows of 64 B = 12 GiB
    const unsigned threads = 1u << 22, iters = 16;
    uint4 *table = nullptr;
    uint32_t *out = nullptr;
    if (hipMalloc(&table, rows * 64) != hipSuccess || hipMalloc(&out, (size_t)(1u << 25) * 4) != hipSuccess) return 1;
    hipMemset(table, 1, rows * 64);
    hipLaunchKernelGGL(k_stream, dim3(threads / 256), dim3(256), 0, 0, table, rows * 4, out); // reference: 12 GiB streamed
    hipLaunchKernelGGL(k_gather, dim3(threads / 256), dim3(256), 0, 0, table, rows, iters, out);
    const unsigned log_n = 26; // 2^26 elements of 32 B = 2 GiB of the table: far beyond the 256 MiB Infinity Cache
    hipLaunchKernelGGL(k_pieces, dim3((1u << log_n) / 1024), dim3(512), 0, 0, table, log_n, out);
    hipDeviceSynchronize();
    printf("k_pieces requested %.3f GB (2^%u elements x 32 B in 128-byte pieces at stride n/256)\n", (double)(1ull << log_n) * 32 / 1e9, log_n);
    printf("k_stream requested %.3f GB; k_gather requested %.3f GB (%u threads x %u rows x 64 B)\n", rows * 64 / 1e9, (double)threads * iters * 64 / 1e9, threads, iters);
    return 0;
}
