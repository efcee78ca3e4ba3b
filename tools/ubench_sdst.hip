// v_mad_u64_u32 issue behaviour on gfx950 (development aid).  The instruction always writes a carry-out SGPR pair (VOP3B sdst); every multiply-add
// of fe29.h names vcc there.  Questions: does a wave issue multiply-adds faster when consecutive ones write DIFFERENT sdst pairs (no write-after-write
// on vcc), and what does a dependent chain (one accumulator, as in a product column) cost against independent ones, at 1 .. 8 waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 ubench_sdst.hip -o bin/ubench_sdst
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
typedef uint64_t u64;
#define ITERS 2048

// MODE 0: 8 independent accumulators, all sdst = vcc
// MODE 1: 8 independent accumulators, sdst rotating over four SGPR pairs
// MODE 2: ONE accumulator (dependent chain of 8 per iteration), sdst = vcc
// MODE 3: ONE accumulator, sdst rotating
// MODE 4: two accumulators interleaved (4 + 4), sdst = vcc
// MODE 5: two accumulators interleaved, sdst rotating
// MODE 6: 8 independent accumulators, v_mul_lo_u32 + v_mul_hi_u32 + 2 adds instead (no sdst on the multiplies; carry through vcc on the adds)
template <int MODE>
__global__ void __launch_bounds__(256) k(u64 *out, u32 a, u32 b)
{
    u64 acc[8];
    u32 x = a + threadIdx.x, y[8];
    for (int c = 0; c < 8; c++) {
        acc[c] = c + threadIdx.x;
        y[c] = b * (c + 3) + blockIdx.x + (threadIdx.x << c);
    }
    for (int it = 0; it < ITERS; it++) {
        if constexpr (MODE == 0) {
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %10, %1\n v_mad_u64_u32 %2, vcc, %8, %11, %2\n v_mad_u64_u32 %3, vcc, %8, %12, %3\n"
                         "v_mad_u64_u32 %4, vcc, %8, %13, %4\n v_mad_u64_u32 %5, vcc, %8, %14, %5\n v_mad_u64_u32 %6, vcc, %8, %15, %6\n v_mad_u64_u32 %7, vcc, %8, %16, %7\n"
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                         : "v"(x), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7])
                         : "vcc");
        } else if constexpr (MODE == 1) {
            asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[22:23], %8, %10, %1\n v_mad_u64_u32 %2, s[24:25], %8, %11, %2\n v_mad_u64_u32 %3, s[26:27], %8, %12, %3\n"
                         "v_mad_u64_u32 %4, s[20:21], %8, %13, %4\n v_mad_u64_u32 %5, s[22:23], %8, %14, %5\n v_mad_u64_u32 %6, s[24:25], %8, %15, %6\n v_mad_u64_u32 %7, s[26:27], %8, %16, %7\n"
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                         : "v"(x), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7])
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if constexpr (MODE == 2) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %3, %0\n v_mad_u64_u32 %0, vcc, %1, %4, %0\n v_mad_u64_u32 %0, vcc, %1, %5, %0\n"
                         "v_mad_u64_u32 %0, vcc, %1, %6, %0\n v_mad_u64_u32 %0, vcc, %1, %7, %0\n v_mad_u64_u32 %0, vcc, %1, %8, %0\n v_mad_u64_u32 %0, vcc, %1, %9, %0\n"
                         : "+v"(acc[0])
                         : "v"(x), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7])
                         : "vcc");
        } else if constexpr (MODE == 3) {
            asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n v_mad_u64_u32 %0, s[22:23], %1, %3, %0\n v_mad_u64_u32 %0, s[24:25], %1, %4, %0\n v_mad_u64_u32 %0, s[26:27], %1, %5, %0\n"
                         "v_mad_u64_u32 %0, s[20:21], %1, %6, %0\n v_mad_u64_u32 %0, s[22:23], %1, %7, %0\n v_mad_u64_u32 %0, s[24:25], %1, %8, %0\n v_mad_u64_u32 %0, s[26:27], %1, %9, %0\n"
                         : "+v"(acc[0])
                         : "v"(x), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7])
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if constexpr (MODE == 4) {
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %4, %1\n v_mad_u64_u32 %0, vcc, %2, %5, %0\n v_mad_u64_u32 %1, vcc, %2, %6, %1\n"
                         "v_mad_u64_u32 %0, vcc, %2, %7, %0\n v_mad_u64_u32 %1, vcc, %2, %8, %1\n v_mad_u64_u32 %0, vcc, %2, %9, %0\n v_mad_u64_u32 %1, vcc, %2, %10, %1\n"
                         : "+v"(acc[0]), "+v"(acc[1])
                         : "v"(x), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7])
                         : "vcc");
        } else if constexpr (MODE == 5) {
            asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n v_mad_u64_u32 %1, s[22:23], %2, %4, %1\n v_mad_u64_u32 %0, s[24:25], %2, %5, %0\n v_mad_u64_u32 %1, s[26:27], %2, %6, %1\n"
                         "v_mad_u64_u32 %0, s[20:21], %2, %7, %0\n v_mad_u64_u32 %1, s[22:23], %2, %8, %1\n v_mad_u64_u32 %0, s[24:25], %2, %9, %0\n v_mad_u64_u32 %1, s[26:27], %2, %10, %1\n"
                         : "+v"(acc[0]), "+v"(acc[1])
                         : "v"(x), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7])
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const u32 lo = x * y[c], hi = __umulhi(x, y[c]);
                acc[c] += ((u64)hi << 32) | lo;
            }
        }
        x ^= (u32)(acc[0] >> 63);
    }
    u64 s = 0;
    for (int c = 0; c < 8; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int blocks, u64 *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    const double ops = (double)ITERS * 8 * blocks * 256;
    const double per_simd_cycle = 256.0 * 4 * 2.4e9; // nominal clock
    printf("%-52s %8.3f ms  %8.2f T mad/s  %5.2f cycles per wave-instruction and SIMD (at 2.4 GHz)\n", name, best, ops / best / 1e9, per_simd_cycle * 64 / (ops / (best * 1e-3)));
}

int main()
{
    u64 *out;
    hipMalloc(&out, (size_t)8192 * 256 * 8);
    for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * wps; // 256 threads = 4 waves = one per SIMD of a CU
        printf("--- %d wave(s) per SIMD ---\n", wps);
        run<0>("8 independent chains, sdst vcc", blocks, out);
        run<1>("8 independent chains, sdst rotating", blocks, out);
        run<2>("1 dependent chain, sdst vcc", blocks, out);
        run<3>("1 dependent chain, sdst rotating", blocks, out);
        run<4>("2 interleaved chains, sdst vcc", blocks, out);
        run<5>("2 interleaved chains, sdst rotating", blocks, out);
        run<6>("8 independent: v_mul_lo + v_mul_hi + add64", blocks, out);
    }
    return 0;
}
