// Structural experiment on the NTT pass: FOUR elements per thread instead of eight.
//
// k_ntt_pass8 (panda_amd/csrc/ntt_radix8.h) keeps 8 elements x 9 limbs per thread: 168 registers, three waves per SIMD, the eight rounds
// of a radix-256 pass as register blocks of 3 + 3 + 2 rounds with two LDS exchanges.  k_ntt_pass4 below is the same pass with 4 elements
// per thread: 256 threads own a tile of 4 sub-transforms (1024 elements), the rounds are four register blocks of 2 rounds with THREE
// exchanges, ~100 registers and five waves per SIMD.  Same butterflies (bfly<> of ntt_radix8.h), same precomputed-quotient products,
// the same number of products per element (13 per 4 elements against 26 per 8; the wave whose twiddles are 1 skips 3 of 13 against
// 7 of 26), per-lane twiddles from the LDS copy of the table in the first two blocks, wave-uniform (scalar) twiddles in the third,
// the 4th root of unity in the last; the output product and its tables as in k_ntt_pass8; bank-conflict-free exchange addresses
// (XOR swizzles, checked with SQ_LDS_BANK_CONFLICT).
//
// This is a TIMING build: the data flow is the real one (every output depends on every input of its sub-transform through real
// butterflies), but the thread-to-element maps were chosen for this experiment and the values are not checked -- like
// tools/ntt8_variants.hip, whose harness this is.  The two kernels run in alternating blocks inside one process.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../panda_amd/csrc ntt4_variant.hip -o bin/ntt4_variant
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "ntt_radix8.h"

#ifndef V4_MINW
#define V4_MINW 5
#endif
#ifndef V4_PB
#define V4_PB 5
#endif
using namespace panda_ntt8;
typedef Bn254Fr F;

constexpr int ELEMS4 = 1024, SUBS4 = 4;

// exchange of 4 elements per thread, PB limb planes at a time (a plane is ELEMS4 words)
template <class Fr, int PB>
__device__ __forceinline__ void exchange4(Fe<Fr> (&e)[4], u32 *s_x, const unsigned (&wa)[4], const unsigned (&ra)[4])
{
    Fe<Fr> n[4];
#pragma unroll
    for (int p0 = 0; p0 < NL; p0 += PB) {
        if (p0 != 0) __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int pl = p0; pl < p0 + PB && pl < NL; pl++) s_x[(pl - p0) * ELEMS4 + wa[m]] = e[m].l[pl];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int pl = p0; pl < p0 + PB && pl < NL; pl++) n[m].l[pl] = s_x[(pl - p0) * ELEMS4 + ra[m]];
    }
#pragma unroll
    for (int m = 0; m < 4; m++) e[m] = n[m];
}

// word of element (s, i) in a plane: the four sub-transforms interleaved, the low three bits of i swizzled by `x` (a function of i's high bits)
__device__ __forceinline__ unsigned word4(unsigned s, unsigned i, unsigned x) { return s | ((i ^ (x & 7u)) << 2); }

template <class Fr, bool FIRST, bool LAST, int PB, int MINW>
__global__ void __launch_bounds__(256, MINW) k_ntt_pass4(Pass8Args A)
{
    constexpr const Plan8<Fr, FIRST ? 2 : 3> &PL = plan8_v<Fr, FIRST ? 2 : 3>;
    __shared__ u32 s_x[PB * ELEMS4];
    __shared__ __attribute__((aligned(16))) u32 s_tw[128 * TW2_STRIDE];
    const unsigned tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned S = (1u << A.log_n) >> 8;
    const unsigned lgp = A.lgp;
    {
        const uint2 *g = reinterpret_cast<const uint2 *>(A.pq) + tid * 5;
        uint2 *l = reinterpret_cast<uint2 *>(s_tw) + tid * 5;
#pragma unroll
        for (int j = 0; j < 5; j++) l[j] = g[j];
    }
    const unsigned blk0 = blockIdx.x * SUBS4;
    const unsigned s = lane & 3, g = lane >> 2; // 16 values of g per wave
    const unsigned j = g | (wave << 4);         // 0 .. 63

    // ---- block 0: i = j + 64 m; rounds 0, 1 (distances 128, 64)
    Fe<Fr> e[4];
    {
        const size_t base = (size_t)(blk0 + s) + (size_t)j * S, step = (size_t)64 * S;
#pragma unroll
        for (int m = 0; m < 4; m++) load_elem32(e[m], A.x + (base + m * step) * 8);
    }
    __syncthreads();
    {
        TwV<Fr> t;
#pragma unroll
        for (int m = 0; m < 2; m++) { // round 0: pairs (m, m + 2), twiddle index j + 64 m
            load_tw2(t, s_tw, j + 64 * m);
            bfly<Fr, PL.b[0], true, PL.red[0], false, 0>(e[m], e[m + 2], t.w, t.q);
        }
        load_tw2(t, s_tw, 2 * j); // round 1: pairs (m, m + 1), twiddle index 2 j
        bfly<Fr, PL.b[1], true, PL.red[1], false, 1>(e[0], e[1], t.w, t.q);
        bfly<Fr, PL.b[1], true, PL.red[1], false, 1>(e[2], e[3], t.w, t.q);
    }
    // ---- exchange 1 -> block 1: i = (j & 15) + 16 m + 64 (j >> 4): in both maps the lanes of a half-wave differ in i[2:0]
    {
        unsigned wa[4], ra[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            wa[m] = word4(s, j + 64 * m, 0);
            ra[m] = word4(s, (j & 15) + 16 * m + 64 * (j >> 4), 0);
        }
        exchange4<Fr, PB>(e, s_x, wa, ra);
    }
    {
        TwV<Fr> t;
#pragma unroll
        for (int m = 0; m < 2; m++) { // round 2 (distance 32): twiddle index 4 ((j & 15) + 16 m)
            load_tw2(t, s_tw, 4 * ((j & 15) + 16 * m));
            bfly<Fr, PL.b[2], true, PL.red[2], false, 0>(e[m], e[m + 2], t.w, t.q);
        }
        load_tw2(t, s_tw, 8 * (j & 15)); // round 3 (distance 16)
        bfly<Fr, PL.b[3], true, PL.red[3], false, 1>(e[0], e[1], t.w, t.q);
        bfly<Fr, PL.b[3], true, PL.red[3], false, 1>(e[2], e[3], t.w, t.q);
    }
    // ---- exchange 2 -> block 2: i = wave + 4 m + 16 g (the wave supplies i[1:0]: the twiddles of rounds 4, 5 are wave-uniform)
    __syncthreads();
    {
        unsigned wa[4], ra[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const unsigned iw = (j & 15) + 16 * m + 64 * (j >> 4), ir = wave + 4 * m + 16 * g;
            wa[m] = word4(s, iw, iw >> 4); // low bits ^ i[6:4]: the writers of a half-wave differ in i[2:0], the readers in i[6:4]
            ra[m] = word4(s, ir, ir >> 4);
        }
        exchange4<Fr, PB>(e, s_x, wa, ra);
    }
    {
        const bool w0 = wave == 0;
        TwV<Fr> t;
#pragma unroll
        for (int m = 0; m < 2; m++) { // round 4 (distance 8): twiddle index 16 (wave + 4 m)
            load_tw2_uniform(t, A.pq, 16 * (wave + 4 * m));
            bfly<Fr, PL.b[4], true, PL.red[4], true, 0>(e[m], e[m + 2], t.w, t.q, m == 0 && w0);
        }
        load_tw2_uniform(t, A.pq, 32 * wave); // round 5 (distance 4)
        bfly<Fr, PL.b[5], true, PL.red[5], true, 1>(e[0], e[1], t.w, t.q, w0);
        bfly<Fr, PL.b[5], true, PL.red[5], true, 1>(e[2], e[3], t.w, t.q, w0);
    }
    // ---- exchange 3 -> block 3: i = m + 4 j
    __syncthreads();
    {
        unsigned wa[4], ra[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const unsigned iw = wave + 4 * m + 16 * g, ir = m + 4 * j;
            // swizzle (i4, i3 ^ i5, i6): the writers of a half-wave differ in i[6:4], the readers in i[4:2]
            wa[m] = word4(s, iw, ((iw >> 4) & 1) | ((((iw >> 3) ^ (iw >> 5)) & 1) << 1) | (((iw >> 6) & 1) << 2));
            ra[m] = word4(s, ir, ((ir >> 4) & 1) | ((((ir >> 3) ^ (ir >> 5)) & 1) << 1) | (((ir >> 6) & 1) << 2));
        }
        exchange4<Fr, PB>(e, s_x, wa, ra);
    }
    {
        TwV<Fr> t;
        load_tw2_uniform(t, A.pq, 64);
        bfly<Fr, PL.b[6], false, PL.red[6], true, 0>(e[0], e[2], nullptr, nullptr); // round 6 (distance 2): 1, and the 4th root of unity
        bfly<Fr, PL.b[6], true, PL.red[6], true, 0>(e[1], e[3], t.w, t.q);
        bfly<Fr, PL.b[7], false, PL.red[7], true, PL.red[7] ? 1 : 2>(e[0], e[1], nullptr, nullptr); // round 7
        bfly<Fr, PL.b[7], false, PL.red[7], true, PL.red[7] ? 1 : 2>(e[2], e[3], nullptr, nullptr);
    }
    constexpr int FB = PL.b[8];
    // ---- output: register m holds output i_out = 64 m' + j' of sub-transform blk (an output map of this experiment: runs of four sub-transforms)
    const unsigned blk = blk0 + s;
    const unsigned p = 1u << lgp, k = blk & (p - 1);
    if constexpr (LAST) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            fe_reduce_mad_2p(e[m]);
            fe_reduce_once(e[m]);
            store_elem32(A.y + ((size_t)blk + ((size_t)(64 * m + j) << lgp)) * 8, e[m]);
        }
    } else {
        const size_t base = ((size_t)(blk - k) << 8) + k + ((size_t)j << lgp);
        const unsigned i2 = (blk >> lgp) >> A.i2_shift;
        if (A.wide) {
            const u32 *row = A.ta + (((size_t)i2 << (lgp + 8)) + k + ((size_t)j << lgp)) * 8;
            uint4 nlo = reinterpret_cast<const uint4 *>(row)[0], nhi = reinterpret_cast<const uint4 *>(row)[1];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const u32 w8[8] = {nlo.x, nlo.y, nlo.z, nlo.w, nhi.x, nhi.y, nhi.z, nhi.w};
                if (m + 1 < 4) {
                    const uint4 *nx = reinterpret_cast<const uint4 *>(row + ((size_t)(64 * (m + 1)) << lgp) * 8);
                    nlo = nx[0];
                    nhi = nx[1];
                }
                Fe<Fr> tw, x, v;
                fe_unpack(tw, w8);
                fe_norm(x, e[m]);
                fe_mul(v, x, tw);
                store_elem32(A.y + (base + ((size_t)(64 * m) << lgp)) * 8, v);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else { // one table (first pass: lgp == 0)
            const unsigned row = i2 << A.ca;
            TwV<Fr> nxt;
            load_tw2(nxt, A.ta, row | j);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const TwV<Fr> t = nxt;
                if (m + 1 < 4) load_tw2(nxt, A.ta, row | (64 * (m + 1) + j));
                Fe<Fr> v;
                fe_mul_shoup<Fr, false>(v, e[m], t.w, t.q);
                store_elem32(A.y + (base + ((size_t)(64 * m) << lgp)) * 8, v);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    (void)FB;
}

int main(int argc, char **argv)
{
    const unsigned log_n = 24; // the pass geometry below (8 + 8 + 8 bits) is that of a 2^24-point transform
    const int rounds = argc > 1 ? atoi(argv[1]) : 6, reps = 10;
    const size_t n = (size_t)1 << log_n;
    u32 *a, *b, *pq, *ta, *wide;
    hipMalloc(&a, n * 32);
    hipMalloc(&b, n * 32);
    hipMalloc(&pq, 128 * TW2_STRIDE * 4);
    hipMalloc(&ta, (size_t)65536 * TW2_STRIDE * 4);
    hipMalloc(&wide, n * 32); // the middle pass's streamed table: 32 bytes per element of the transform
    {
        std::vector<u32> h(n * 8);
        u32 x = 12345;
        for (size_t i = 0; i < h.size(); i++) {
            x = x * 1664525u + 1013904223u;
            h[i] = (i & 7) == 7 ? (x >> 4) : x; // below 2^252 < p
        }
        hipMemcpy(a, h.data(), n * 32, hipMemcpyHostToDevice);
        hipMemcpy(wide, h.data(), n * 32, hipMemcpyHostToDevice);
        std::vector<u32> t((size_t)65536 * TW2_STRIDE);
        for (size_t i = 0; i < t.size(); i++) {
            x = x * 1664525u + 1013904223u;
            t[i] = x & LIMB_MASK;
        }
        hipMemcpy(ta, t.data(), t.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(pq, t.data(), 128 * TW2_STRIDE * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t e[4];
    for (auto &ev : e) hipEventCreate(&ev);
    Pass8Args p1{}, p2{}, p3{};
    p1.x = a; p1.y = b; p1.pq = pq; p1.ta = ta; p1.tb = ta; p1.log_n = log_n; p1.lgp = 0; p1.ca = 8; p1.cb = 0; p1.i2_shift = log_n - 16;
    p2 = p1; p2.x = b; p2.y = a; p2.lgp = 8; p2.ca = 8; p2.cb = 8; p2.i2_shift = log_n - 24; p2.wide = 1; p2.ta = wide; // the shipped middle pass: streamed table
    p3 = p1; p3.lgp = 16; p3.ca = p3.cb = 0;
    std::vector<float> med[2][4];
    for (int r = 0; r < rounds; r++) {
        for (int which = 0; which < 2; which++) {
            const int v = (r & 1) ? 1 - which : which; // A B / B A / A B ...
            std::vector<float> t[4];
            for (int i = 0; i < reps + 2; i++) {
                hipEventRecord(e[0]);
                if (v == 0) {
                    p1.tiles = p2.tiles = p3.tiles = (unsigned)(n / ELEMS);
                    hipLaunchKernelGGL((k_ntt_pass8<F, true, false, 5, 3>), dim3((unsigned)(n / ELEMS)), dim3(THREADS), 0, 0, p1);
                    hipEventRecord(e[1]);
                    hipLaunchKernelGGL((k_ntt_pass8<F, false, false, 5, 3>), dim3((unsigned)(n / ELEMS)), dim3(THREADS), 0, 0, p2);
                    hipEventRecord(e[2]);
                    hipLaunchKernelGGL((k_ntt_pass8<F, false, true, 5, 3>), dim3((unsigned)(n / ELEMS)), dim3(THREADS), 0, 0, p3);
                } else {
                    hipLaunchKernelGGL((k_ntt_pass4<F, true, false, V4_PB, V4_MINW>), dim3((unsigned)(n / ELEMS4)), dim3(256), 0, 0, p1);
                    hipEventRecord(e[1]);
                    hipLaunchKernelGGL((k_ntt_pass4<F, false, false, V4_PB, V4_MINW>), dim3((unsigned)(n / ELEMS4)), dim3(256), 0, 0, p2);
                    hipEventRecord(e[2]);
                    hipLaunchKernelGGL((k_ntt_pass4<F, false, true, V4_PB, V4_MINW>), dim3((unsigned)(n / ELEMS4)), dim3(256), 0, 0, p3);
                }
                hipEventRecord(e[3]);
                hipEventSynchronize(e[3]);
                if (i < 2) continue;
                float ms;
                for (int k = 0; k < 3; k++) {
                    hipEventElapsedTime(&ms, e[k], e[k + 1]);
                    t[k].push_back(ms);
                }
                hipEventElapsedTime(&ms, e[0], e[3]);
                t[3].push_back(ms);
            }
            for (int k = 0; k < 4; k++) {
                std::sort(t[k].begin(), t[k].end());
                med[v][k].push_back(t[k][t[k].size() / 2]);
            }
        }
    }
    for (int v = 0; v < 2; v++) {
        float m[4];
        for (int k = 0; k < 4; k++) {
            std::sort(med[v][k].begin(), med[v][k].end());
            m[k] = med[v][k][med[v][k].size() / 2];
        }
        printf("%-58s 2^%u: first %.3f  middle %.3f  last %.3f  total %.3f ms  (median of %d block medians, blocks alternating)\n",
               v == 0 ? "k_ntt_pass8 (8 elements / thread, PB 5, 3 waves / SIMD)" : "k_ntt_pass4 (4 elements / thread, 3 exchanges)", log_n, m[0], m[1], m[2], m[3], rounds);
    }
    printf("k_ntt_pass4 build: PB=%d, __launch_bounds__(256, %d); %s\n", V4_PB, V4_MINW, hipGetErrorString(hipGetLastError()));
    return 0;
}
