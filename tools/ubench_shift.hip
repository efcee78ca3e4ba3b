// Micro-benchmark: the carry step between two columns of a product-scanning multiplication -- acc >>= 29 on a 64-bit accumulator -- as
// one v_lshrrev_b64 (what hipcc emits) against v_alignbit_b32 + v_lshrrev_b32, alone and inside a column (nine v_mad_u64_u32, mask, shift).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 ubench_shift.hip -o bin/ubench_shift
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
typedef uint64_t u64;
#define ITERS 2048
#define CH 8

__device__ __forceinline__ u64 shr29_b64(u64 a)
{
    u64 r;
    asm volatile("v_lshrrev_b64 %0, 29, %1" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ u64 shr29_pair(u64 a)
{
    u32 lo = (u32)a, hi = (u32)(a >> 32), nlo, nhi;
    asm volatile("v_alignbit_b32 %0, %2, %3, 29\n\tv_lshrrev_b32 %1, 29, %2" : "=&v"(nlo), "=v"(nhi) : "v"(hi), "v"(lo));
    return ((u64)nhi << 32) | nlo;
}

template <int V>
__global__ void __launch_bounds__(256) k_shift(u64 *out, u32 a)
{
    u64 acc[CH];
    for (int c = 0; c < CH; c++) acc[c] = ((u64)(a + c) << 40) + threadIdx.x * 0x9E3779B97F4A7C15ull;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            acc[c] = V == 0 ? shr29_b64(acc[c]) : shr29_pair(acc[c]);
            acc[c] |= 0x8000000000000000ull >> (it & 7);
        }
    }
    u64 s = 0;
    for (int c = 0; c < CH; c++) s ^= acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// a column: nine multiply-adds into the accumulator, low limb masked off, accumulator shifted
template <int V>
__global__ void __launch_bounds__(256) k_column(u64 *out, u32 a)
{
    u32 x[9], y[9];
    for (int i = 0; i < 9; i++) {
        x[i] = (a * (i + 3) + threadIdx.x) & 0x1fffffff;
        y[i] = (a * (i + 7) + blockIdx.x) & 0x1fffffff;
    }
    u64 acc = threadIdx.x;
    u32 sum = 0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int col = 0; col < 4; col++) {
            asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\tv_mad_u64_u32 %0, vcc, %11, %12, %0\n\tv_mad_u64_u32 %0, vcc, %13, %14, %0\n\tv_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
                "v_mad_u64_u32 %0, vcc, %17, %18, %0"
                : "+v"(acc)
                : "v"(x[0]), "v"(y[col]), "v"(x[1]), "v"(y[col + 1]), "v"(x[2]), "v"(y[col + 2]), "v"(x[3]), "v"(y[col + 3]), "v"(x[4]), "v"(y[col + 4]), "v"(x[5]),
                  "v"(y[(col + 5) % 9]), "v"(x[6]), "v"(y[(col + 6) % 9]), "v"(x[7]), "v"(y[(col + 7) % 9]), "v"(x[8]), "v"(y[(col + 8) % 9])
                : "vcc");
            sum += (u32)acc & 0x1fffffff;
            acc = V == 0 ? shr29_b64(acc) : shr29_pair(acc);
        }
        x[it & 7] ^= sum & 0xff;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + sum;
}

template <class K>
static float time_kernel(K launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main()
{
    u64 *out;
    (void)hipMalloc(&out, 8 * 256 * 8192);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        printf("--- %d waves/SIMD ---\n", wps);
        float ms = time_kernel([&] { k_shift<0><<<blocks, 256>>>(out, 7); });
        printf("v_lshrrev_b64 (+ v_or)              %8.3f ms  %8.2f G shifts/s\n", ms, (double)blocks * 256 * ITERS * CH / (ms * 1e-3) * 1e-9);
        ms = time_kernel([&] { k_shift<1><<<blocks, 256>>>(out, 7); });
        printf("v_alignbit_b32 + v_lshrrev_b32 (+or) %8.3f ms  %8.2f G shifts/s\n", ms, (double)blocks * 256 * ITERS * CH / (ms * 1e-3) * 1e-9);
        ms = time_kernel([&] { k_column<0><<<blocks, 256>>>(out, 7); });
        printf("column (9 mads), 64-bit shift        %8.3f ms  %8.2f G columns/s\n", ms, (double)blocks * 256 * ITERS * 4 / (ms * 1e-3) * 1e-9);
        ms = time_kernel([&] { k_column<1><<<blocks, 256>>>(out, 7); });
        printf("column (9 mads), alignbit pair       %8.3f ms  %8.2f G columns/s\n", ms, (double)blocks * 256 * ITERS * 4 / (ms * 1e-3) * 1e-9);
    }
    return 0;
}
