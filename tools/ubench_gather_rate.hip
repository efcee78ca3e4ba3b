// ubench_gather_rate.hip -- how many random 64-byte rows per second the memory system of an MI355X delivers, by table size and by
// resident waves per SIMD: the ceiling k_accumulate's row gathers (one per addition, 12 GiB of tables at 2^24 points) sit under.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_gather_rate tools/ubench_gather_rate.hip ; run: ./gpurun_out/ubench_gather_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int ILP>
__global__ void __launch_bounds__(128) k_gather(const uint4 *__restrict__ table, uint64_t rows, unsigned iters, uint32_t *__restrict__ out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = t * 0x9E3779B97F4A7C15ull + 12345;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (unsigned i = 0; i < iters; i += ILP) {
        uint4 v[ILP][4];
#pragma unroll
        for (int j = 0; j < ILP; j++) { // ILP independent rows in flight per lane
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const uint4 *p = table + ((s >> 20) % rows) * 4;
            v[j][0] = p[0]; v[j][1] = p[1]; v[j][2] = p[2]; v[j][3] = p[3];
        }
#pragma unroll
        for (int j = 0; j < ILP; j++) {
            acc.x ^= v[j][0].x ^ v[j][1].y ^ v[j][2].z ^ v[j][3].w;
            acc.y += v[j][0].y + v[j][1].z + v[j][2].w + v[j][3].x;
        }
    }
    out[t] = acc.x ^ acc.y;
}

// the same rows with four lanes per row: instruction m of 4 reads the rows of lanes 16m .. 16m+15, piece (lane & 3) each, so one
// instruction touches 16 rows (16 pages) instead of 64; the pieces go through LDS back to the lane that owns the row.
__global__ void __launch_bounds__(128) k_gather_coop(const uint4 *__restrict__ table, uint64_t rows, unsigned iters, uint32_t *__restrict__ out)
{
    __shared__ uint4 stage[128 * 4];
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t s = t * 0x9E3779B97F4A7C15ull + 12345;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 *mine = stage + wave * 256;
    for (unsigned i = 0; i < iters; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t row = (s >> 20) % rows;
        uint4 v[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const unsigned src = 16 * m + (lane >> 2);
            const uint64_t r = ((uint64_t)__shfl((unsigned)(row >> 32), src) << 32) | __shfl((unsigned)row, src);
            v[m] = table[r * 4 + (lane & 3)];
        }
#pragma unroll
        for (int m = 0; m < 4; m++) mine[m * 64 + lane] = v[m]; // row of lane 16m + lane / 4, piece lane & 3: row-major
        __builtin_amdgcn_s_waitcnt(0xc07f);
        uint4 w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = mine[lane * 4 + j];
        acc.x ^= w[0].x ^ w[1].y ^ w[2].z ^ w[3].w;
        acc.y += w[0].y + w[1].z + w[2].w + w[3].x;
    }
    out[t] = acc.x ^ acc.y;
}

int main(int argc, char **argv)
{
    uint32_t *out;
    uint4 *table;
    const uint64_t max_rows = (12ull << 30) / 64;
    // argv[1] = 1: physically contiguous memory (hipExtMallocWithFlags, hipDeviceMallocContiguous): larger translation fragments
    const bool contiguous = argc > 1 && argv[1][0] == '1';
    hipError_t ae = contiguous ? hipExtMallocWithFlags((void **)&table, max_rows * 64, hipDeviceMallocContiguous) : hipMalloc(&table, max_rows * 64);
    printf("allocation: %s -> %s\n", contiguous ? "hipDeviceMallocContiguous" : "hipMalloc", hipGetErrorName(ae));
    if (ae != hipSuccess) return 1;
    hipMalloc(&out, (1u << 22) * 4);
    hipMemset(table, 1, max_rows * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("table_GiB,waves_per_simd,rows_in_flight_per_lane,G_rows_per_s,GB_per_s\n");
    for (double gib : {0.125, 0.5, 1.0, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0})
        for (unsigned waves : {4u})
            for (int ilp : {1, 0}) { // 0: four lanes per row
                const uint64_t rows = (uint64_t)(gib * (1ull << 30)) / 64;
                const unsigned threads = 256 * 4 * 64 * waves, iters = 256;
                auto launch = [&] {
                    if (ilp == 0)
                        hipLaunchKernelGGL(k_gather_coop, dim3(threads / 128), dim3(128), 0, 0, table, rows, iters, out);
                    else if (ilp == 1)
                        hipLaunchKernelGGL(k_gather<1>, dim3(threads / 128), dim3(128), 0, 0, table, rows, iters, out);
                    else
                        hipLaunchKernelGGL(k_gather<4>, dim3(threads / 128), dim3(128), 0, 0, table, rows, iters, out);
                };
                launch();
                hipEventRecord(e0, 0);
                for (int r = 0; r < 5; r++) launch();
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double rps = 5.0 * threads * iters / (ms * 1e-3);
                printf("%.3f,%u,%d,%.2f,%.0f\n", gib, waves, ilp, rps / 1e9, rps * 64 / 1e9);
            }
    return 0;
}
