// Micro-benchmark: fe29 multiply variants under a madd-like dependent workload, at several occupancies.
//   hipcc --offload-arch=gfx950 -O3 -I../panda_amd/csrc ubench_mul.hip -o ubench_mul
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "fe29.h"
#include "fe29_chain.h"
using namespace panda29;
typedef Bn254Fq F;

__device__ __forceinline__ void mad_vv(u64 &acc, u32 a, u32 b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
__device__ __forceinline__ void mad_vs(u64 &acc, u32 a, u32 b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b) : "vcc"); }

// serial-chain variant: every product accumulates straight into the running column accumulator
__device__ __forceinline__ void fe_mul_asm(Fe<F> &r, const Fe<F> &a, const Fe<F> &b)
{
    constexpr int N = F::N;
    u32 m[N], out[N];
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mad_vv(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mad_vs(acc, m[i], F::P[k - i]);
        m[k] = ((u32)acc * F::INV) & LIMB_MASK;
        mad_vs(acc, m[k], F::P[0]);
        acc >>= LIMB_BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) mad_vv(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - N + 1; i < N; i++) mad_vs(acc, m[i], F::P[k - i]);
        out[k - N] = (u32)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
    }
    out[N - 1] = (u32)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
}

// column-block variant: the multiply-adds of a column in one or two asm blocks (MacChain), no s_nop between them
__device__ __forceinline__ void fe_mul_blk(Fe<F> &r, const Fe<F> &a, const Fe<F> &b)
{
    constexpr int N = F::N;
    u32 m[N], out[N];
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        switch (k) { // constant after unrolling
        case 0: MacChain<1>::vv(acc, &a.l[0], &b.l[0]); break;
        case 1: MacChain<2>::vv(acc, &a.l[0], &b.l[1]); MacChain<1>::vs(acc, &m[0], &F::P[1]); break;
        case 2: MacChain<3>::vv(acc, &a.l[0], &b.l[2]); MacChain<2>::vs(acc, &m[0], &F::P[2]); break;
        case 3: MacChain<4>::vv(acc, &a.l[0], &b.l[3]); MacChain<3>::vs(acc, &m[0], &F::P[3]); break;
        case 4: MacChain<5>::vv(acc, &a.l[0], &b.l[4]); MacChain<4>::vs(acc, &m[0], &F::P[4]); break;
        case 5: MacChain<6>::vv(acc, &a.l[0], &b.l[5]); MacChain<5>::vs(acc, &m[0], &F::P[5]); break;
        case 6: MacChain<7>::vv(acc, &a.l[0], &b.l[6]); MacChain<6>::vs(acc, &m[0], &F::P[6]); break;
        case 7: MacChain<8>::vv(acc, &a.l[0], &b.l[7]); MacChain<7>::vs(acc, &m[0], &F::P[7]); break;
        default: MacChain<9>::vv(acc, &a.l[0], &b.l[8]); MacChain<8>::vs(acc, &m[0], &F::P[8]); break;
        }
        m[k] = ((u32)acc * F::INV) & LIMB_MASK;
        mad_vs(acc, m[k], F::P[0]);
        acc >>= LIMB_BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
        switch (k) {
        case 9: MacChain<8>::vv(acc, &a.l[1], &b.l[8]); MacChain<8>::vs(acc, &m[1], &F::P[8]); break;
        case 10: MacChain<7>::vv(acc, &a.l[2], &b.l[8]); MacChain<7>::vs(acc, &m[2], &F::P[8]); break;
        case 11: MacChain<6>::vv(acc, &a.l[3], &b.l[8]); MacChain<6>::vs(acc, &m[3], &F::P[8]); break;
        case 12: MacChain<5>::vv(acc, &a.l[4], &b.l[8]); MacChain<5>::vs(acc, &m[4], &F::P[8]); break;
        case 13: MacChain<4>::vv(acc, &a.l[5], &b.l[8]); MacChain<4>::vs(acc, &m[5], &F::P[8]); break;
        case 14: MacChain<3>::vv(acc, &a.l[6], &b.l[8]); MacChain<3>::vs(acc, &m[6], &F::P[8]); break;
        case 15: MacChain<2>::vv(acc, &a.l[7], &b.l[8]); MacChain<2>::vs(acc, &m[7], &F::P[8]); break;
        default: MacChain<1>::vv(acc, &a.l[8], &b.l[8]); MacChain<1>::vs(acc, &m[8], &F::P[8]); break;
        }
        out[k - N] = (u32)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
    }
    out[N - 1] = (u32)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
}

#define ITERS 256
template <int VARIANT>
__global__ void __launch_bounds__(256) k_mul(u32 *out, const u32 *in)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fe<F> x, y, z, w;
    for (int j = 0; j < 9; j++) {
        x.l[j] = in[(i & 1023) * 18 + j] & LIMB_MASK;
        y.l[j] = in[(i & 1023) * 18 + 9 + j] & LIMB_MASK;
    }
    x.l[8] &= 0x3fffff; y.l[8] &= 0x3fffff;
    z = y; w = x; z.l[0] ^= 5; w.l[1] ^= 9;
    for (int it = 0; it < ITERS; it++) { // two independent chains, like the independent products inside a madd
        if (VARIANT == 0) { fe_mul(x, x, y); fe_mul(z, z, w); fe_mul(y, y, x); fe_mul(w, w, z); }
        else if (VARIANT == 1) { fe_mul_asm(x, x, y); fe_mul_asm(z, z, w); fe_mul_asm(y, y, x); fe_mul_asm(w, w, z); }
        else { fe_mul_blk(x, x, y); fe_mul_blk(z, z, w); fe_mul_blk(y, y, x); fe_mul_blk(w, w, z); }
    }
    u32 s = 0;
    for (int j = 0; j < 9; j++) s += x.l[j] * 3 + y.l[j] * 5 + z.l[j] * 7 + w.l[j] * 11;
    out[i] = s;
}

template <int V>
static void run(const char *name, int blocks, u32 *out, u32 *in, u32 *chk)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_mul<V><<<blocks, 256>>>(out, in);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k_mul<V><<<blocks, 256>>>(out, in);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    double muls = (double)blocks * 256 * ITERS * 4;
    hipMemcpy(chk, out, 4, hipMemcpyDeviceToHost);
    printf("%-10s blocks=%5d  %8.3f ms  %8.2f G mulmod/s  (check %08x)\n", name, blocks, ms, muls / (ms * 1e-3) * 1e-9, *chk);
}

int main()
{
    u32 *out, *in; hipMalloc(&out, 4 * 256 * 8192); hipMalloc(&in, 1024 * 18 * 4);
    std::vector<u32> h(1024 * 18); for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2654435761u + 12345u);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    u32 chk;
    for (int wps : {2, 3, 4, 8}) {
        int blocks = 256 * wps;
        printf("--- %d waves/SIMD ---\n", wps);
        run<0>("compiler", blocks, out, in, &chk);
        run<1>("asm-chain", blocks, out, in, &chk);
        run<2>("asm-block", blocks, out, in, &chk);
    }
    return 0;
}
