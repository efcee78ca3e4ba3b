// Micro-benchmark: is a block-Karatsuba product worth it now that plain VALU instructions are known to issue in the shadow of the
// quarter-rate multiply-adds?  9 limbs = 3 blocks of 3: six 3x3 block products (54 v_mad_u64_u32) and ~100 plain 32/64-bit operations
// instead of 81 multiply-adds, followed by the usual interleaved Montgomery reduction (81 + 9).  Compared with fe_mul (162 + 9) in the
// same dependent-chain harness as tools/ubench_mul.hip.
//   hipcc --offload-arch=gfx950 -O3 -I../panda_amd/csrc ubench_karatsuba.hip -o ubench_karatsuba
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "fe29.h"
using namespace panda29;
typedef Bn254Fq F;

// 3 x 3 limbs -> 5 columns
__device__ __forceinline__ void blk3(u64 (&c)[5], const u32 *a, const u32 *b)
{
#pragma unroll
    for (int j = 0; j < 5; j++) c[j] = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    MacChain<1>::vv(c[0], a, b);
    MacChain<2>::vv(c[1], a, b + 1);
    MacChain<3>::vv(c[2], a, b + 2);
    MacChain<2>::vv(c[3], a + 1, b + 2);
    MacChain<1>::vv(c[4], a + 2, b + 2);
#endif
}

// limbs of a, b below 2^30
__device__ __forceinline__ void fe_mul_kara(Fe<F> &r, const Fe<F> &a, const Fe<F> &b)
{
    constexpr int N = 9;
    u32 sa01[3], sa02[3], sa12[3], sb01[3], sb02[3], sb12[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        sa01[i] = a.l[i] + a.l[3 + i];
        sa02[i] = a.l[i] + a.l[6 + i];
        sa12[i] = a.l[3 + i] + a.l[6 + i];
        sb01[i] = b.l[i] + b.l[3 + i];
        sb02[i] = b.l[i] + b.l[6 + i];
        sb12[i] = b.l[3 + i] + b.l[6 + i];
    }
    u64 p0[5], p1[5], p2[5], p01[5], p02[5], p12[5];
    blk3(p0, a.l, b.l);
    blk3(p1, a.l + 3, b.l + 3);
    blk3(p2, a.l + 6, b.l + 6);
    blk3(p01, sa01, sb01);
    blk3(p02, sa02, sb02);
    blk3(p12, sa12, sb12);
    u64 col[17];
#pragma unroll
    for (int k = 0; k < 17; k++) col[k] = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        col[j] += p0[j];
        col[3 + j] += p01[j] - p0[j] - p1[j];
        col[6 + j] += p02[j] - p0[j] - p2[j] + p1[j];
        col[9 + j] += p12[j] - p1[j] - p2[j];
        col[12 + j] += p2[j];
    }
    // interleaved Montgomery reduction over the product columns
    u32 m[N], out[N];
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        acc += col[k];
#pragma unroll
        for (int i = 0; i < k; i++) FE29_MAC_CONST(acc, m[i], F::P[k - i]);
        m[k] = ((u32)acc * F::INV) & LIMB_MASK;
        FE29_MAC_CONST(acc, m[k], F::P[0]);
        acc >>= LIMB_BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
        acc += col[k];
#pragma unroll
        for (int i = k - N + 1; i < N; i++) FE29_MAC_CONST(acc, m[i], F::P[k - i]);
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
    x.l[8] &= 0x3fffff;
    y.l[8] &= 0x3fffff;
    z = y;
    w = x;
    z.l[0] ^= 5;
    w.l[1] ^= 9;
    for (int it = 0; it < ITERS; it++) { // two independent chains, like the independent products inside a mixed addition
        if (VARIANT == 0) {
            fe_mul(x, x, y);
            fe_mul(z, z, w);
            fe_mul(y, y, x);
            fe_mul(w, w, z);
        } else {
            fe_mul_kara(x, x, y);
            fe_mul_kara(z, z, w);
            fe_mul_kara(y, y, x);
            fe_mul_kara(w, w, z);
        }
    }
    u32 s = 0;
    for (int j = 0; j < 9; j++) s += x.l[j] * 3 + y.l[j] * 5 + z.l[j] * 7 + w.l[j] * 11;
    out[i] = s;
}

template <int V>
static void run(const char *name, int blocks, u32 *out, u32 *in)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_mul<V><<<blocks, 256>>>(out, in);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k_mul<V><<<blocks, 256>>>(out, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    double muls = (double)blocks * 256 * ITERS * 4;
    u32 chk[4];
    hipMemcpy(chk, out, 16, hipMemcpyDeviceToHost);
    printf("%-12s blocks=%5d  %8.3f ms  %8.2f G mulmod/s  (check %08x %08x)\n", name, blocks, ms, muls / (ms * 1e-3) * 1e-9, chk[0], chk[3]);
}

int main()
{
    u32 *out, *in;
    hipMalloc(&out, 4 * 256 * 8192);
    hipMalloc(&in, 1024 * 18 * 4);
    std::vector<u32> h(1024 * 18);
    for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2654435761u + 12345u);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int wps : {1, 2, 3, 4, 8}) {
        int blocks = 256 * wps;
        printf("--- %d waves/SIMD ---\n", wps);
        run<0>("fe_mul", blocks, out, in);
        run<1>("karatsuba", blocks, out, in);
    }
    return 0;
}
