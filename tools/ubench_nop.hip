// ubench_nop.hip -- does the s_nop the compiler puts after every inline-asm v_mad_u64_u32 cost anything?
// A: 16 dependent multiply-adds per loop iteration as 16 asm statements (s_nop 0 after each, as in fe29.h);
// B: the same 16 in ONE asm block (no s_nop inside).  4 waves per SIMD, like k_accumulate.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_nop tools/ubench_nop.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define MAD(acc, x, y) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc")

__global__ void __launch_bounds__(256) k_separate(uint64_t *out, uint32_t a, uint32_t b, unsigned iters)
{
    uint64_t acc = threadIdx.x;
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (unsigned i = 0; i < iters; i++) {
        MAD(acc, x, y); MAD(acc, x, y); MAD(acc, x, y); MAD(acc, x, y);
        MAD(acc, x, y); MAD(acc, x, y); MAD(acc, x, y); MAD(acc, x, y);
        MAD(acc, x, y); MAD(acc, x, y); MAD(acc, x, y); MAD(acc, x, y);
        MAD(acc, x, y); MAD(acc, x, y); MAD(acc, x, y); MAD(acc, x, y);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_block(uint64_t *out, uint32_t a, uint32_t b, unsigned iters)
{
    uint64_t acc = threadIdx.x;
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (unsigned i = 0; i < iters; i++) {
        asm volatile(
            "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
            "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
            "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
            "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0"
            : "+v"(acc) : "v"(x), "v"(y) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    const unsigned blocks = 256 * 4, iters = 20000; // 4 waves per SIMD on 256 CUs
    uint64_t *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int which = 0; which < 2; which++)
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL(k_separate, dim3(blocks), dim3(256), 0, 0, out, 123u, 456u, iters);
            else hipLaunchKernelGGL(k_block, dim3(blocks), dim3(256), 0, 0, out, 123u, 456u, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s rep %d: %.3f ms, %.1f G mad/s\n", which ? "one asm block  " : "16 asm stmts   ", rep, ms, (double)blocks * 256 * iters * 16 / ms / 1e6);
        }
    return 0;
}
