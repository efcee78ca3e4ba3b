// Which XCD does workgroup w of a 1-D grid run on?  (ChunkGeom assumes w mod 8; only performance depends on it.)  A grid shaped like
// k_accumulate's (12288 workgroups of 128 threads, ~100 registers' worth of occupancy emulated by LDS) records HW_REG_XCC_ID per workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_map.hip -o tools/bin/xcd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void __launch_bounds__(128) k_map(unsigned *out, unsigned spin)
{
    __shared__ unsigned pad[4096]; // 16 KB per workgroup, like the accumulate with its sector buffers
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned v = threadIdx.x;
    for (unsigned i = 0; i < spin; i++) v = v * 1664525u + 1013904223u; // live for a while, so that later workgroups are placed as slots free up
    pad[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 15u) | (pad[5] & 0x100u ? 0u : 0u);
}
int main()
{
    const unsigned wgs = 12288;
    unsigned *d;
    hipMalloc(&d, wgs * 4);
    for (unsigned spin : {0u, 20000u, 200000u}) {
        hipMemset(d, 0xff, wgs * 4);
        k_map<<<wgs, 128>>>(d, spin);
        hipDeviceSynchronize();
        std::vector<unsigned> h(wgs);
        hipMemcpy(h.data(), d, wgs * 4, hipMemcpyDeviceToHost);
        unsigned match = 0, hist[8][8] = {};
        for (unsigned w = 0; w < wgs; w++) {
            match += h[w] == (w & 7u);
            hist[w & 7u][h[w] & 7u]++;
        }
        printf("spin %6u: %u of %u workgroups ran on XCD (w mod 8)\n", spin, match, wgs);
        for (unsigned r = 0; r < 8; r++) {
            printf("   w mod 8 = %u ->", r);
            for (unsigned x = 0; x < 8; x++) printf(" %5u", hist[r][x]);
            printf("\n");
        }
    }
    return 0;
}
