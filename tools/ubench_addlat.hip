// Micro-benchmark: latency of a chain of dependent XYZZ additions / doublings for a wave that runs ALONE on its SIMD -- the regime of the
// fix-up and bucket-reduction trees behind k_accumulate -- in four spellings: one lane per addition with product-scanning products
// (what the library ran until round 3), the same with operand-scanning products (Lat<> below), four lanes per addition (curve29_quad.h)
// with either product.  Every variant's sum is compared with the first on the host (projective equality).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../panda_amd/csrc ubench_addlat.hip -o bin/ubench_addlat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "curve29_quad.h"
using namespace panda29;
typedef Bn254Fq F;
// Operand-scanning products (row i adds a_i * b_j into 2N - 1 separate column accumulators: the multiply-adds of a row are independent,
// where fe_mul's product scanning makes every column one dependent chain).  Same multiply-add count and column sums; measured here
// because a lone wave issues DEPENDENT multiply-adds at 41 % of the instruction's rate (profiles/r01_ubench_int_rates.txt).  It does
// not help: hipcc already interleaves the independent products of an addition, and a lone wave is issue-bound (see the output).
template <class B>
struct Lat : B {
};
template <class B>
PANDA_HD void rows_mac(u64 *t, const u32 *a, const u32 *b)
{
#pragma unroll
    for (int i = 0; i < B::N; i++)
#pragma unroll
        for (int j = 0; j < B::N; j++) t[i + j] += (u64)a[i] * b[j];
}
template <class B>
PANDA_HD void rows_reduce(u32 *out, u64 *t)
{
    constexpr int N = B::N;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const u32 m = ((u32)t[i] * B::INV) & LIMB_MASK;
#pragma unroll
        for (int j = 0; j < N; j++) t[i + j] += (u64)m * B::P[j];
        t[i + 1] += t[i] >> LIMB_BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
        out[k - N] = (u32)t[k] & LIMB_MASK;
        t[k + 1] += t[k] >> LIMB_BITS;
    }
    out[N - 1] = (u32)t[2 * N - 1];
}
namespace panda29 {
template <class B>
PANDA_HD void fe_mul(Fe<Lat<B>> &r, const Fe<Lat<B>> &a, const Fe<Lat<B>> &b)
{
    u64 t[2 * B::N] = {0};
    rows_mac<B>(t, a.l, b.l);
    rows_reduce<B>(r.l, t);
}
template <class B>
PANDA_HD void fe_sqr(Fe<Lat<B>> &r, const Fe<Lat<B>> &a) { fe_mul(r, a, a); }
template <class B>
PANDA_HD void fe_mul_add(Fe<Lat<B>> &r, const Fe<Lat<B>> &a, const Fe<Lat<B>> &b, const Fe<Lat<B>> &c, const Fe<Lat<B>> &d)
{
    u64 t[2 * B::N] = {0};
    rows_mac<B>(t, a.l, b.l);
    rows_mac<B>(t, c.l, d.l);
    rows_reduce<B>(r.l, t);
}
} // namespace panda29
typedef Lat<F> FL;

constexpr int TBL = 64;
constexpr int PW = 4 * F::N;

__global__ void k_table(u32 *tbl)
{
    if (threadIdx.x) return;
    Fe<F> gx, gy;
    fe_from_u32(gx, 1);
    fe_from_u32(gy, 2);
    Xyzz<F> acc;
    xyzz_from_affine(acc, gx, gy);
    for (int j = 0; j < TBL; j++) {
        if (j) xyzz_madd(acc, gx, gy, false);
        for (int i = 0; i < F::N; i++) {
            tbl[j * PW + i] = acc.X.l[i];
            tbl[j * PW + F::N + i] = acc.Y.l[i];
            tbl[j * PW + 2 * F::N + i] = acc.ZZ.l[i];
            tbl[j * PW + 3 * F::N + i] = acc.ZZZ.l[i];
        }
    }
}

template <class FF>
__device__ __forceinline__ void load_pt(Xyzz<FF> &p, const u32 *src)
{
    for (int i = 0; i < F::N; i++) {
        p.X.l[i] = src[i];
        p.Y.l[i] = src[F::N + i];
        p.ZZ.l[i] = src[2 * F::N + i];
        p.ZZZ.l[i] = src[3 * F::N + i];
    }
}
template <class FF>
__device__ __forceinline__ void store_pt(u32 *dst, const Xyzz<FF> &p)
{
    for (int i = 0; i < F::N; i++) {
        dst[i] = p.X.l[i];
        dst[F::N + i] = p.Y.l[i];
        dst[2 * F::N + i] = p.ZZ.l[i];
        dst[3 * F::N + i] = p.ZZZ.l[i];
    }
}

// VARIANT 0 / 1: one lane per chain (plain / Lat);  2 / 3: one quad per chain (plain / Lat).  DBL: every fourth step doubles instead.
template <int VARIANT, bool DBL>
__global__ void __launch_bounds__(64) k_chain(u32 *out, const u32 *tbl, int iters)
{
    typedef typename std::conditional<(VARIANT & 1) != 0, FL, F>::type FF;
    constexpr bool QUAD = VARIANT >= 2;
    const unsigned lane = threadIdx.x, role = lane & 3u;
    const unsigned chain = QUAD ? (blockIdx.x * 16 + (lane >> 2)) : (blockIdx.x * 64 + lane);
    Xyzz<FF> acc, q, d;
    load_pt(acc, tbl + (chain % TBL) * PW);
    for (int it = 0; it < iters; it++) {
        load_pt(q, tbl + ((chain * 7 + it * 3 + 1) % TBL) * PW);
        if (DBL && (it & 3) == 3) {
            if constexpr (QUAD) xyzz_dbl_quad(d, acc, role);
            else xyzz_dbl(d, acc);
            acc = d;
        } else {
            if constexpr (QUAD) xyzz_add_quad(acc, q, role);
            else xyzz_add(acc, q);
        }
    }
    if (!QUAD || role == 0) store_pt(out + (size_t)chain * PW, acc);
}

static bool same_point(const u32 *a, const u32 *b)
{
    Xyzz<F> p, q;
    for (int i = 0; i < F::N; i++) {
        p.X.l[i] = a[i]; p.Y.l[i] = a[F::N + i]; p.ZZ.l[i] = a[2 * F::N + i]; p.ZZZ.l[i] = a[3 * F::N + i];
        q.X.l[i] = b[i]; q.Y.l[i] = b[F::N + i]; q.ZZ.l[i] = b[2 * F::N + i]; q.ZZZ.l[i] = b[3 * F::N + i];
    }
    if (xyzz_is_identity(p) || xyzz_is_identity(q)) return xyzz_is_identity(p) && xyzz_is_identity(q);
    Fe<F> l, r, d;
    fe_mul(l, p.X, q.ZZ); fe_mul(r, q.X, p.ZZ); fe_sub<F, 2>(d, l, r);
    if (!fe_is_zero_mod_p(d)) return false;
    fe_mul(l, p.Y, q.ZZZ); fe_mul(r, q.Y, p.ZZZ); fe_sub<F, 2>(d, l, r);
    return fe_is_zero_mod_p(d);
}

template <int V, bool DBL>
static void run(const char *name, int waves, int iters, u32 *out, const u32 *tbl, std::vector<u32> &ref, int chains_checked)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_chain<V, DBL><<<waves, 64>>>(out, tbl, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k_chain<V, DBL><<<waves, 64>>>(out, tbl, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    std::vector<u32> h((size_t)chains_checked * PW);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    if (ref.empty()) ref = h;
    else for (int c = 0; c < chains_checked; c++) bad += !same_point(h.data() + (size_t)c * PW, ref.data() + (size_t)c * PW);
    printf("%-34s waves=%5d  %8.3f ms  %7.2f us per step  mismatches=%d\n", name, waves, ms, ms * 1e3 / iters, bad);
}

int main()
{
    u32 *out, *tbl;
    hipMalloc(&out, (size_t)8192 * 64 * PW * 4);
    hipMalloc(&tbl, TBL * PW * 4);
    k_table<<<1, 64>>>(tbl);
    hipDeviceSynchronize();
    const int iters = 64;
    for (int waves : {256, 1024, 2048, 4096}) { // 1 wave per CU, 1 / 2 / 4 per SIMD
        printf("--- %d waves (%.2f per SIMD) ---\n", waves, waves / 1024.0);
        std::vector<u32> ref, refd;
        run<0, false>("lane, product scanning", waves, iters, out, tbl, ref, 16);
        run<1, false>("lane, operand scanning", waves, iters, out, tbl, ref, 16);
        run<2, false>("quad, product scanning", waves, iters, out, tbl, ref, 16);
        run<3, false>("quad, operand scanning", waves, iters, out, tbl, ref, 16);
        run<0, true>("lane, product scanning, 1/4 dbl", waves, iters, out, tbl, refd, 16);
        run<1, true>("lane, operand scanning, 1/4 dbl", waves, iters, out, tbl, refd, 16);
        run<2, true>("quad, product scanning, 1/4 dbl", waves, iters, out, tbl, refd, 16);
        run<3, true>("quad, operand scanning, 1/4 dbl", waves, iters, out, tbl, refd, 16);
    }
    return 0;
}
