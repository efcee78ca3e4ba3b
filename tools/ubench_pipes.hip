// Do v_mad_u64_u32 and v_fma_f64 share an execution pipe on gfx950 (development aid)?  Eight independent chains per lane: all integer multiply-adds, all
// double-precision FMAs, or four of each interleaved.  If the mixed kernel sustains more operations per second than either pure one, the two instructions
// issue to different units and a modular product could spread its columns over both.
//   hipcc -w --offload-arch=gfx950 -O3 ubench_pipes.hip -o bin/ubench_pipes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
typedef uint64_t u64;
#define ITERS 2048

template <int MODE> // 0: 8 mad, 1: 8 dfma, 2: 4 mad + 4 dfma interleaved, 3: 4 mad only, 4: 4 dfma only
__global__ void __launch_bounds__(256) k(u64 *out, u32 a, double fa)
{
    u64 acc[8];
    double d[8];
    u32 x = a + threadIdx.x;
    double fx = fa + threadIdx.x * 1e-9;
    for (int c = 0; c < 8; c++) {
        acc[c] = c + threadIdx.x;
        d[c] = 1.0 + c * 1e-3 + threadIdx.x * 1e-6;
    }
    for (int it = 0; it < ITERS; it++) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 8; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"((u32)(0x12345 + c)) : "vcc");
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 8; c++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[c]) : "v"(fx), "v"(d[(c + 1) & 7]));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"((u32)(0x12345 + c)) : "vcc");
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[c]) : "v"(fx), "v"(d[(c + 1) & 3]));
            }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int c = 0; c < 4; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"((u32)(0x12345 + c)) : "vcc");
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[c]) : "v"(fx), "v"(d[(c + 1) & 3]));
        }
    }
    u64 s = 0;
    double ds = 0;
    for (int c = 0; c < 8; c++) {
        s += acc[c];
        ds += d[c];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (u64)ds;
}

template <int MODE>
void run(const char *name, int blocks, u64 *out, int ops_per_iter)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 3u, 1.000001);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    const double ops = (double)ITERS * ops_per_iter * blocks * 256;
    printf("%-44s %8.3f ms  %8.2f T ops/s\n", name, best, ops / best / 1e9);
}

int main()
{
    u64 *out;
    hipMalloc(&out, (size_t)8192 * 256 * 8);
    for (int wps : {2, 4, 8}) {
        const int blocks = 256 * wps;
        printf("--- %d wave(s) per SIMD ---\n", wps);
        run<0>("8 x v_mad_u64_u32", blocks, out, 8);
        run<1>("8 x v_fma_f64", blocks, out, 8);
        run<2>("4 x v_mad_u64_u32 + 4 x v_fma_f64 interleaved", blocks, out, 8);
        run<3>("4 x v_mad_u64_u32", blocks, out, 4);
        run<4>("4 x v_fma_f64", blocks, out, 4);
    }
    return 0;
}
