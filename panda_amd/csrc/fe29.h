// fe29.h -- prime-field arithmetic on 29-bit limbs for gfx950 (and, unchanged, for the host).
//
// Replaces the reference's device field layer (src/cuda/core/field/field.cuh:139-486: 32-bit-limb
// even/odd CIOS built on PTX mad.lo.cc/madc.hi.cc carry chains, asm/ptx.cuh) with a design
// that fits CDNA4's integer pipe:
//
//   * gfx950 has no carry-chained multiply-add; its wide multiply is v_mad_u64_u32
//     (32x32 + 64 -> 64, half rate, measured 46 lane-ops/clk/CU, profiles/r01_ubench_int_rates.txt).
//   * so elements are held as N limbs of 29 bits in 32-bit registers ("unsaturated"): a column of
//     the schoolbook product, N products of < 2^58 plus the N Montgomery products m_i*p_j, fits a
//     64-bit accumulator without any carry handling.  A multiply is then 2*N^2 back-to-back
//     v_mad_u64_u32 plus N (v_mul_lo, v_and) and 2N shifts: 925 cycles per wave for N = 9 against
//     1700 for the 8x32 CIOS the compiler can build from the same instruction set.
//   * Montgomery radix is R = 2^(29 N) (BN254: 2^261, R/p = 169).  The spare factor is spent on lazy
//     reduction: products come back in [0, 2p); sums and differences are not reduced at all, only
//     carry-normalised, as long as the bounds below hold.
//
// The wire format stays the reference's (Montgomery form with R_wire = 2^(32 L), 32-bit limbs,
// field_storage.cuh:12-16); fe_from_wire / fe_to_wire convert with one extra multiply.
//
// Bounds contract (N = 9 figures; checked exhaustively by tests/host_check/fe29_host.cpp through tests/test_fe29_host.py):
//   limb classes   tight : limbs 0..N-2 <  2^29             (outputs of fe_mul / fe_sqr / fe_unpack)
//                  loose : limbs 0..N-2 <= 2^29 + 8         (outputs of fe_norm / fe_sub / fe_add)
//                  raw   : limbs 0..N-2 <  2^30 + 16        (fe_add_nr of two loose values)
//   fe_mul(a, b)   needs  limb(a) < 2^30.5, limb(b) < 2^30  and  value(a) * value(b) < 0.9 R p
//                  gives  tight, value < 2p  (exactly: < a b / R + p)
//   fe_sub<KB>     needs  value(b) < KB p, limb(b) < 2^31 - 4; gives loose, value < value(a) + KEFF p
#pragma once
#include <stdint.h>

#include "fe29_params.h"

#if defined(__HIPCC__)
#define PANDA_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define PANDA_HD inline __attribute__((always_inline))
#endif

#if defined(FE29_CHECK)
#include <assert.h>
#define FE29_SHADOW_DECL unsigned __int128 shadow = 0;
#define FE29_SHADOW_MAC(x, y) shadow += (unsigned __int128)(x) * (y);
#define FE29_SHADOW_SHIFT()                                      \
    assert(shadow < ((unsigned __int128)1 << 64) && "fe29 column accumulator overflow"); \
    shadow >>= 29;
#else
#define FE29_SHADOW_DECL
#define FE29_SHADOW_MAC(x, y)
#define FE29_SHADOW_SHIFT()
#endif

// Multiply-accumulate into a 64-bit column.  On the device this is spelled as v_mad_u64_u32 directly: left to itself
// hipcc sums each column in a scratch accumulator and merges it with a separate 64-bit add (v_lshl_add_u64, half
// rate), one extra instruction per column; the explicit chain is 7-11 % faster (tools/ubench_mul.hip).  The carry
// output of the instruction is unused (columns cannot overflow, see the bounds contract) and lands in VCC.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FE29_NO_ASM)
#define FE29_MAC(acc, x, y) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc")
#define FE29_MAC_CONST(acc, x, c) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(c) : "vcc")
#else
#define FE29_MAC(acc, x, y) acc += (u64)(x) * (y)
#define FE29_MAC_CONST(acc, x, c) acc += (u64)(x) * (c)
#endif

#if defined(__HIP_DEVICE_COMPILE__) && !defined(FE29_NO_ASM)
#define FE29_DEVICE_CHAINS 1
#include <utility>

#include "fe29_chain.h"
#endif

namespace panda29 {

typedef uint32_t u32;
typedef uint64_t u64;
constexpr u32 LIMB_BITS = 29;
constexpr u32 LIMB_MASK = (1u << 29) - 1;

template <class F>
struct Fe {
    u32 l[F::N];
};

// quadratic extension B[u] / (u^2 + 1) as a "field descriptor" (fe29_ext2.h): an element is c0's limbs followed by c1's.
// Functions that are called with explicit template arguments (fe_sub, fe_neg) dispatch on IsExt2 themselves; the rest is overloaded.
template <class B>
struct Ext2;
template <class F>
struct IsExt2 {
    static constexpr bool value = false;
};
template <class B>
struct IsExt2<Ext2<B>> {
    static constexpr bool value = true;
};
template <class B>
PANDA_HD Fe<B> &ext_c0(Fe<Ext2<B>> &a) { return *reinterpret_cast<Fe<B> *>(&a.l[0]); }
template <class B>
PANDA_HD Fe<B> &ext_c1(Fe<Ext2<B>> &a) { return *reinterpret_cast<Fe<B> *>(&a.l[B::N]); }
template <class B>
PANDA_HD const Fe<B> &ext_c0(const Fe<Ext2<B>> &a) { return *reinterpret_cast<const Fe<B> *>(&a.l[0]); }
template <class B>
PANDA_HD const Fe<B> &ext_c1(const Fe<Ext2<B>> &a) { return *reinterpret_cast<const Fe<B> *>(&a.l[B::N]); }

template <class F>
PANDA_HD void fe_zero(Fe<F> &r)
{
#pragma unroll
    for (int i = 0; i < F::N; i++) r.l[i] = 0;
}

template <class F>
PANDA_HD void fe_one(Fe<F> &r)
{
#pragma unroll
    for (int i = 0; i < F::N; i++) r.l[i] = F::ONE[i];
}

template <class F>
PANDA_HD void fe_const(Fe<F> &r, const u32 (&c)[F::N])
{
#pragma unroll
    for (int i = 0; i < F::N; i++) r.l[i] = c[i];
}

// all limbs zero (exact representation test, not a congruence test)
template <class F>
PANDA_HD bool fe_all_zero(const Fe<F> &a)
{
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) acc |= a.l[i];
    return acc == 0;
}

#if defined(FE29_DEVICE_CHAINS)
// Device spelling of the product-scanning columns: the multiply-adds of a column go out as two or three asm blocks
// (MacChain, fe29_chain.h) instead of one asm statement each, which keeps hipcc from separating them with s_nop:
// +3.5 % multiplications/s at four waves per SIMD, +15 % at three (tools/ubench_mul.hip).
template <class F, int K>
__device__ inline __attribute__((always_inline)) void fe29_reduce_col(uint64_t &acc, uint32_t *m, uint32_t *out)
{
    constexpr int N = F::N;
    // A modulus whose top limb is zero (BLS12-377 Fq: 377 bits = 13 x 29 exactly, held in 14 limbs for the lazy-reduction headroom) has one
    // product m_i * P[N-1] = 0 in each of the columns N-1 .. 2N-2: they are not issued (14 of the 392 multiply-adds of a product).
    constexpr bool TOP0 = F::P[N - 1] == 0;
    if constexpr (K < N) {
        if constexpr (TOP0 && K == N - 1)
            MacChain<K - 1>::vs(acc, m + 1, &F::P[K - 1]);
        else
            MacChain<K>::vs(acc, m, &F::P[K]);
        m[K] = ((uint32_t)acc * F::INV) & ((1u << 29) - 1);
        MacChain<1>::vs(acc, m + K, &F::P[0]);
    } else {
        if constexpr (TOP0)
            MacChain<2 * N - 2 - K>::vs(acc, m + (K - N + 2), &F::P[N - 2]);
        else
            MacChain<2 * N - 1 - K>::vs(acc, m + (K - N + 1), &F::P[N - 1]);
        out[K - N] = (uint32_t)acc & ((1u << 29) - 1);
    }
    acc >>= 29;
}
template <class F, int K>
__device__ inline __attribute__((always_inline)) void fe29_mul_col(uint64_t &acc, const uint32_t *a, const uint32_t *b, uint32_t *m, uint32_t *out)
{
    constexpr int N = F::N;
    if constexpr (K == 0)
        MacChain<1>::vv0(acc, a, b); // column 0 starts the accumulator (no register pair to clear)
    else if constexpr (K < N)
        MacChain<K + 1>::vv(acc, a, b + K);
    else
        MacChain<2 * N - 1 - K>::vv(acc, a + (K - N + 1), b + (N - 1));
    fe29_reduce_col<F, K>(acc, m, out);
}
template <class F, int K>
__device__ inline __attribute__((always_inline)) void fe29_sqr_col(uint64_t &acc, const uint32_t *a, const uint32_t *a2, uint32_t *m, uint32_t *out)
{
    constexpr int N = F::N;
    constexpr int I0 = K < N ? 0 : K - N + 1;
    constexpr int CNT = (K + 1) / 2 - I0; // cross terms a2[i] * a[K - i], I0 <= i, 2 i < K
    MacChain<CNT>::vv(acc, a2 + I0, a + (K - I0));
    if constexpr (K == 0)
        MacChain<1>::vv0(acc, a, a);
    else if constexpr ((K & 1) == 0)
        MacChain<1>::vv(acc, a + K / 2, a + K / 2);
    fe29_reduce_col<F, K>(acc, m, out);
}
template <class F, int... K>
__device__ inline __attribute__((always_inline)) void fe29_mul_cols(uint64_t &acc, const uint32_t *a, const uint32_t *b, uint32_t *m, uint32_t *out, std::integer_sequence<int, K...>)
{
    (fe29_mul_col<F, K>(acc, a, b, m, out), ...);
}
template <class F, int K>
__device__ inline __attribute__((always_inline)) void fe29_mul_add_col(uint64_t &acc, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d, uint32_t *m,
                                                                       uint32_t *out)
{
    constexpr int N = F::N;
    if constexpr (K == 0) {
        MacChain<1>::vv0(acc, a, b);
        MacChain<1>::vv(acc, c, d);
    } else if constexpr (K < N) {
        MacChain<K + 1>::vv(acc, a, b + K);
        MacChain<K + 1>::vv(acc, c, d + K);
    } else {
        MacChain<2 * N - 1 - K>::vv(acc, a + (K - N + 1), b + (N - 1));
        MacChain<2 * N - 1 - K>::vv(acc, c + (K - N + 1), d + (N - 1));
    }
    fe29_reduce_col<F, K>(acc, m, out);
}
template <class F, int... K>
__device__ inline __attribute__((always_inline)) void fe29_mul_add_cols(uint64_t &acc, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d,
                                                                        uint32_t *m, uint32_t *out, std::integer_sequence<int, K...>)
{
    (fe29_mul_add_col<F, K>(acc, a, b, c, d, m, out), ...);
}
template <class F, int... K>
__device__ inline __attribute__((always_inline)) void fe29_sqr_cols(uint64_t &acc, const uint32_t *a, const uint32_t *a2, uint32_t *m, uint32_t *out, std::integer_sequence<int, K...>)
{
    (fe29_sqr_col<F, K>(acc, a, a2, m, out), ...);
}
#endif

// r = a * b / R mod p.  Product scanning: column k of a*b and of m*p accumulate in one u64.
template <class F>
PANDA_HD void fe_mul(Fe<F> &r, const Fe<F> &a, const Fe<F> &b)
{
    constexpr int N = F::N;
    u32 m[N];
    u32 out[N];
    u64 acc = 0;
#if defined(FE29_DEVICE_CHAINS)
    fe29_mul_cols<F>(acc, a.l, b.l, m, out, std::make_integer_sequence<int, 2 * N - 1>());
    out[N - 1] = (u32)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
    return;
#endif
    FE29_SHADOW_DECL
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            FE29_MAC(acc, a.l[i], b.l[k - i]);
            FE29_SHADOW_MAC(a.l[i], b.l[k - i])
        }
#pragma unroll
        for (int i = 0; i < k; i++) {
            FE29_MAC_CONST(acc, m[i], F::P[k - i]);
            FE29_SHADOW_MAC(m[i], F::P[k - i])
        }
        m[k] = ((u32)acc * F::INV) & LIMB_MASK;
        FE29_MAC_CONST(acc, m[k], F::P[0]);
        FE29_SHADOW_MAC(m[k], F::P[0])
        acc >>= LIMB_BITS;
        FE29_SHADOW_SHIFT()
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            FE29_MAC(acc, a.l[i], b.l[k - i]);
            FE29_SHADOW_MAC(a.l[i], b.l[k - i])
        }
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            FE29_MAC_CONST(acc, m[i], F::P[k - i]);
            FE29_SHADOW_MAC(m[i], F::P[k - i])
        }
        out[k - N] = (u32)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
        FE29_SHADOW_SHIFT()
    }
    out[N - 1] = (u32)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
}

// r = (a*b + c*d) / R mod p with ONE Montgomery reduction: both products accumulate into the same columns
// (27 terms of < 2^58 fit a u64 for N = 9; N = 14 needs all four operands tight or loose).  Saves N^2 of the
// 4 N^2 multiply-adds two separate fe_mul would spend.  value(a b + c d) < 0.9 R p; result tight, < 2p.
template <class F>
PANDA_HD void fe_mul_add(Fe<F> &r, const Fe<F> &a, const Fe<F> &b, const Fe<F> &c, const Fe<F> &d)
{
    constexpr int N = F::N;
    static_assert(3 * N * (1ull << 58) + (3ull * N << 38) < (1ull << 63) * 2 - 1, "column accumulator too narrow");
    u32 m[N];
    u32 out[N];
    u64 acc = 0;
#if defined(FE29_DEVICE_CHAINS)
    fe29_mul_add_cols<F>(acc, a.l, b.l, c.l, d.l, m, out, std::make_integer_sequence<int, 2 * N - 1>());
    out[N - 1] = (u32)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
    return;
#endif
    FE29_SHADOW_DECL
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            FE29_MAC(acc, a.l[i], b.l[k - i]);
            FE29_SHADOW_MAC(a.l[i], b.l[k - i])
            FE29_MAC(acc, c.l[i], d.l[k - i]);
            FE29_SHADOW_MAC(c.l[i], d.l[k - i])
        }
#pragma unroll
        for (int i = 0; i < k; i++) {
            FE29_MAC_CONST(acc, m[i], F::P[k - i]);
            FE29_SHADOW_MAC(m[i], F::P[k - i])
        }
        m[k] = ((u32)acc * F::INV) & LIMB_MASK;
        FE29_MAC_CONST(acc, m[k], F::P[0]);
        FE29_SHADOW_MAC(m[k], F::P[0])
        acc >>= LIMB_BITS;
        FE29_SHADOW_SHIFT()
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            FE29_MAC(acc, a.l[i], b.l[k - i]);
            FE29_SHADOW_MAC(a.l[i], b.l[k - i])
            FE29_MAC(acc, c.l[i], d.l[k - i]);
            FE29_SHADOW_MAC(c.l[i], d.l[k - i])
        }
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            FE29_MAC_CONST(acc, m[i], F::P[k - i]);
            FE29_SHADOW_MAC(m[i], F::P[k - i])
        }
        out[k - N] = (u32)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
        FE29_SHADOW_SHIFT()
    }
    out[N - 1] = (u32)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
}

// r = a^2 / R mod p; cross terms once, against the doubled operand.  limb(a) < 2^30.
template <class F>
PANDA_HD void fe_sqr(Fe<F> &r, const Fe<F> &a)
{
    constexpr int N = F::N;
    u32 m[N], out[N], a2[N];
#pragma unroll
    for (int i = 0; i < N; i++) a2[i] = a.l[i] << 1;
    u64 acc = 0;
#if defined(FE29_DEVICE_CHAINS)
    fe29_sqr_cols<F>(acc, a.l, a2, m, out, std::make_integer_sequence<int, 2 * N - 1>());
    out[N - 1] = (u32)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
    return;
#endif
    FE29_SHADOW_DECL
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) {
            FE29_MAC(acc, a2[i], a.l[k - i]);
            FE29_SHADOW_MAC(a2[i], a.l[k - i])
        }
        if ((k & 1) == 0) {
            FE29_MAC(acc, a.l[k / 2], a.l[k / 2]);
            FE29_SHADOW_MAC(a.l[k / 2], a.l[k / 2])
        }
#pragma unroll
        for (int i = 0; i < k; i++) {
            FE29_MAC_CONST(acc, m[i], F::P[k - i]);
            FE29_SHADOW_MAC(m[i], F::P[k - i])
        }
        m[k] = ((u32)acc * F::INV) & LIMB_MASK;
        FE29_MAC_CONST(acc, m[k], F::P[0]);
        FE29_SHADOW_MAC(m[k], F::P[0])
        acc >>= LIMB_BITS;
        FE29_SHADOW_SHIFT()
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; 2 * i < k; i++) {
            FE29_MAC(acc, a2[i], a.l[k - i]);
            FE29_SHADOW_MAC(a2[i], a.l[k - i])
        }
        if ((k & 1) == 0) {
            FE29_MAC(acc, a.l[k / 2], a.l[k / 2]);
            FE29_SHADOW_MAC(a.l[k / 2], a.l[k / 2])
        }
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            FE29_MAC_CONST(acc, m[i], F::P[k - i]);
            FE29_SHADOW_MAC(m[i], F::P[k - i])
        }
        out[k - N] = (u32)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
        FE29_SHADOW_SHIFT()
    }
    out[N - 1] = (u32)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
}

// one round of parallel carries: limbs < 2^32 in, loose out.  Value unchanged.
template <class F>
PANDA_HD void fe_norm(Fe<F> &r, const Fe<F> &t)
{
    constexpr int N = F::N;
    u32 c[N];
#pragma unroll
    for (int i = 0; i < N - 1; i++) c[i] = t.l[i] >> LIMB_BITS;
    u32 o[N];
    o[0] = t.l[0] & LIMB_MASK;
#pragma unroll
    for (int i = 1; i < N - 1; i++) o[i] = (t.l[i] & LIMB_MASK) + c[i - 1];
    o[N - 1] = t.l[N - 1] + c[N - 2];
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = o[i];
}

// limb-wise sum, no carries ("raw")
template <class F>
PANDA_HD void fe_add_nr(Fe<F> &r, const Fe<F> &a, const Fe<F> &b)
{
#pragma unroll
    for (int i = 0; i < F::N; i++) r.l[i] = a.l[i] + b.l[i];
}

template <class F>
PANDA_HD void fe_add(Fe<F> &r, const Fe<F> &a, const Fe<F> &b)
{
    Fe<F> t;
    fe_add_nr(t, a, b);
    fe_norm(r, t);
}

// extra multiples of p a subtraction must add so that its top limb cannot underflow
template <class F>
struct SubMargin {
    // top limb of p in units of 1: BN254-like fields have ~2^21 there, the 14-limb BLS12-377 Fq < 1
    static constexpr int value = (F::P[F::N - 1] >= 16) ? 1 : 6;
};

// r = a - b + KEFF*p, b < KB*p.  Result loose.
template <class F, int KB>
PANDA_HD void fe_sub(Fe<F> &r, const Fe<F> &a, const Fe<F> &b)
{
    if constexpr (IsExt2<F>::value) {
        fe_sub<typename F::Base, KB>(ext_c0(r), ext_c0(a), ext_c0(b));
        fe_sub<typename F::Base, KB>(ext_c1(r), ext_c1(a), ext_c1(b));
        return;
    } else {
    constexpr int K = KB + SubMargin<F>::value;
    static_assert(K <= 200, "subtraction constant table too small");
    Fe<F> t;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
#if defined(FE29_CHECK)
        assert((u64)a.l[i] + F::KP[K][i] >= b.l[i] && (u64)a.l[i] + F::KP[K][i] - b.l[i] < (1ull << 32) && "fe_sub limb range");
#endif
        t.l[i] = a.l[i] + F::KP[K][i] - b.l[i];
    }
    fe_norm(r, t);
    }
}

// KP[K] with the per-limb bias lowered from 2^31 to 2^30 (same value): for differences that feed a multiplication
// without being normalised first
template <class F, int K>
PANDA_HD constexpr u32 fe_kp30(int i)
{
    return i == 0 ? F::KP[K][0] - (2u << LIMB_BITS) : (i < F::N - 1 ? F::KP[K][i] - (2u << LIMB_BITS) + 2u : F::KP[K][i] + 2u);
}

// fields whose column accumulators have room for one un-normalised operand (limbs < 2^31) in fe_mul / fe_mul_add
template <class F>
struct RawOperandOk {
    static constexpr bool value = F::N <= 9;
};

// r = a - b + KEFF*p, b < KB*p, WITHOUT the carry pass: limbs < 2^31 ("raw31").  a, b tight or loose.
// The result may only be ONE operand of a fe_mul whose other operand is tight or loose, or the last operand of
// fe_mul_add; it must not be squared, stored, or fed to another addition.  Saves the 3N instructions of fe_norm.
template <class F, int KB>
PANDA_HD void fe_sub_raw(Fe<F> &r, const Fe<F> &a, const Fe<F> &b)
{
    static_assert(RawOperandOk<F>::value, "no column headroom for a raw operand in this field");
    constexpr int K = KB + SubMargin<F>::value;
    static_assert(K <= 200, "subtraction constant table too small");
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        const u32 kp = fe_kp30<F, K>(i);
#if defined(FE29_CHECK)
        assert((u64)a.l[i] + kp >= b.l[i] && (u64)a.l[i] + kp - b.l[i] < (1ull << 31) + (1ull << 24) && "fe_sub_raw limb range");
#endif
        r.l[i] = a.l[i] + kp - b.l[i];
    }
}

// The same with the per-limb bias at U * 2^29 (U = 2 is fe_sub_raw, U = 3 admits an un-normalised b: limbs <= 2^30 + 16, the sum of two
// loose values).  a, b limbs <= 2^30 + 16; the result has limbs < 3 * 2^30 + 32 and may only be the x operand of fe_mul_shoup (whose
// columns have room for exactly that) or the input of a reduction.
template <class F, int K, int U>
PANDA_HD constexpr u32 fe_kp_bias(int i)
{
    static_assert(U >= 2 && U <= 4, "bias in units of 2^29");
    return i == 0 ? F::KP[K][0] - ((u32)(4 - U) << LIMB_BITS) : (i < F::N - 1 ? F::KP[K][i] - ((u32)(4 - U) << LIMB_BITS) + (u32)(4 - U) : F::KP[K][i] + (u32)(4 - U));
}
template <class F, int KB, int U>
PANDA_HD void fe_sub_raw_bias(Fe<F> &r, const Fe<F> &a, const Fe<F> &b)
{
    static_assert(RawOperandOk<F>::value, "no column headroom for a raw operand in this field");
    constexpr int K = KB + SubMargin<F>::value;
    static_assert(K <= 200, "subtraction constant table too small");
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        const u32 kp = fe_kp_bias<F, K, U>(i);
#if defined(FE29_CHECK)
        assert((u64)a.l[i] + kp >= b.l[i] && (u64)a.l[i] + kp - b.l[i] < 3 * (1ull << 30) + 64 && "fe_sub_raw_bias limb range");
#endif
        r.l[i] = a.l[i] + kp - b.l[i];
    }
}

// r = KEFF*p - a, a < KB*p, without the carry pass (see fe_sub_raw)
template <class F, int KB>
PANDA_HD void fe_neg_raw(Fe<F> &r, const Fe<F> &a)
{
    static_assert(RawOperandOk<F>::value, "no column headroom for a raw operand in this field");
    constexpr int K = KB + SubMargin<F>::value;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        const u32 kp = fe_kp30<F, K>(i);
#if defined(FE29_CHECK)
        assert(kp >= a.l[i] && "fe_neg_raw limb range");
#endif
        r.l[i] = kp - a.l[i];
    }
}

// value bound added by fe_sub<F,KB>, in units of p
template <class F, int KB>
struct SubGrowth {
    static constexpr int value = F::KEFF[KB + SubMargin<F>::value];
};

// r = KEFF*p - a  (negation), a < KB*p
template <class F, int KB>
PANDA_HD void fe_neg(Fe<F> &r, const Fe<F> &a)
{
    if constexpr (IsExt2<F>::value) {
        fe_neg<typename F::Base, KB>(ext_c0(r), ext_c0(a));
        fe_neg<typename F::Base, KB>(ext_c1(r), ext_c1(a));
        return;
    } else {
    constexpr int K = KB + SubMargin<F>::value;
    Fe<F> t;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
#if defined(FE29_CHECK)
        assert(F::KP[K][i] >= a.l[i] && "fe_neg limb range");
#endif
        t.l[i] = F::KP[K][i] - a.l[i];
    }
    fe_norm(r, t);
    }
}

// 32-bit wire limbs -> 29-bit limbs (same integer); tight
template <class F>
PANDA_HD void fe_unpack(Fe<F> &r, const u32 *w)
{
    constexpr int N = F::N, L = F::L;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int bit = 29 * k, wi = bit >> 5, sh = bit & 31;
        u32 v = 0;
        if (wi < L) {
            v = w[wi] >> sh;
            if (sh > 3 && wi + 1 < L) v |= w[wi + 1] << (32 - sh);
        }
        r.l[k] = (k < N - 1) ? (v & LIMB_MASK) : v;
    }
}

// 29-bit limbs -> 32-bit wire limbs; input must be canonical-tight (all limbs < 2^29, value < 2^(32 L))
template <class F>
PANDA_HD void fe_pack(u32 *w, const Fe<F> &a)
{
    constexpr int N = F::N, L = F::L;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int bit = 32 * i, j = bit / 29, s = bit - 29 * j;
        u32 v = a.l[j] >> s;
        if (j + 1 < N) v |= a.l[j + 1] << (29 - s);
        if (29 - s + 29 < 32 && j + 2 < N) v |= a.l[j + 2] << (58 - s);
        w[i] = v;
    }
}

// sequential carry propagation: any limbs < 2^32 -> tight (value unchanged)
template <class F>
PANDA_HD void fe_carry(Fe<F> &a)
{
#pragma unroll
    for (int i = 0; i < F::N - 1; i++) {
        a.l[i + 1] += a.l[i] >> LIMB_BITS;
        a.l[i] &= LIMB_MASK;
    }
}

// tight value in [0, 2p) -> canonical [0, p)
template <class F>
PANDA_HD void fe_reduce_once(Fe<F> &a)
{
    constexpr int N = F::N;
    u32 d[N];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u32 t = a.l[i] - F::P[i] - borrow;
        borrow = t >> 31;
        d[i] = (i < N - 1) ? (t & LIMB_MASK) : t;
    }
#pragma unroll
    for (int i = 0; i < N; i++) a.l[i] = borrow ? a.l[i] : d[i];
}

// limbs < 2^32, value < 2^9 p  ->  canonical [0, p), without a multiply.
// The quotient is estimated from the top limb with a 2^-52 fixed-point reciprocal of (top limb of p) + 1,
// which never overshoots and undershoots by at most one; needs a field whose p has a wide top limb.
template <class F>
PANDA_HD void fe_reduce_small_2p(Fe<F> &a) // the same down to [0, 2p), tight: enough wherever another reduction follows
{
    constexpr int N = F::N;
    static_assert(F::P[N - 1] >= (1u << 16), "fe_reduce_small: top limb of p too narrow for the quotient estimate");
    fe_carry(a);
    constexpr u64 C = (1ull << 52) / ((u64)F::P[N - 1] + 1);
    const u32 m = (u32)(((u64)a.l[N - 1] * C) >> 52);
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        int64_t t = (int64_t)a.l[i] - (int64_t)((u64)m * F::P[i]) + carry;
        a.l[i] = (i < N - 1) ? ((u32)t & LIMB_MASK) : (u32)t;
        carry = t >> LIMB_BITS;
    }
}

template <class F>
PANDA_HD void fe_reduce_small(Fe<F> &a)
{
    fe_reduce_small_2p(a);
    fe_reduce_once(a);
}

// ------------------------------------------------------------------ products by constants with a precomputed quotient
//
// For a constant w in [0, p) keep wq = floor(w R / p) next to it (R = 2^(29 N)).  For any x below 8 R the estimate
// q = floor(x wq / R) satisfies  x w / p - x / R - 1 < q <= x w / p,  so  x w - q p  is x w mod p up to (x / R + 2) multiples
// of p -- and, being below R, it is determined by the LOW halves of x w and q p alone.  Against the Montgomery product
// (2 N^2 multiply-adds, N v_mul_lo for the quotient digits) this is N(N+1)/2 for x w, as many for q p and the top
// N(N-1)/2 + 2N - 1 of x wq (two guard columns below R keep the truncation error of q under one unit): 143 instead of 162 + 9
// for N = 9.  No radix factor appears: x w mod p is in whatever form x was (the NTT multiplies wire-form residues by
// plain-integer twiddles and gets wire-form residues back).
//
//   fe_mul_shoup   needs  limb(x) < 3 * 2^30 + 64 (tight, loose, or the output of fe_sub_raw / fe_sub_raw_bias: nine such limbs times a
//                         29-bit limb of the constant, plus the nine q * (R - p) terms of the low half, stay below 2^64), value(x) < R for the bound given
//                  gives  tight, value < 3p  (from the estimate above: < (x / R + 2) p)
// The bound that callers which PACK the result into 32 L bits rely on is tighter.  The truncated quotient product differs from
// Q = floor(x wq / R) only by a borrow out of the dropped columns, which are worth less than N 2^-29 < 2^-23 of a unit of q: the computed
// q is Q - 1 only when frac(x wq / R) < 2^-23, and Q otherwise.  Either way q > x wq / R - 1 - 2^-23 >= x w / p - x / R - 1 - 2^-23, so
//                         value < (1 + x / R + 2^-23) p
// -- below 2^(32 L) for every field here as long as x / R stays below about one (BLS12-381 Fr, p = 0.453 2^256, is the tightest:
// k_ntt_pass8 asserts it for the bound its plan gives the operand).
template <class F>
struct FeTw {
    u32 w[F::N];  // the constant, canonical
    u32 q[F::N];  // floor(w R / p)
};

#if defined(FE29_DEVICE_CHAINS)
template <class F, bool UNIFORM, int C>
__device__ inline __attribute__((always_inline)) void fe29_shoup_hi_col(uint64_t &acc, const uint32_t *x, const uint32_t *wq, uint32_t *q)
{
    constexpr int N = F::N;
    constexpr int I0 = C < N ? 0 : C - N + 1;
    constexpr int CNT = (C < N ? C : N - 1) - I0 + 1;
    if constexpr (C == N - 2) { // the first column of the quotient product starts the accumulator
        if constexpr (UNIFORM)
            MacChain<CNT>::vs0(acc, x + I0, wq + (C - I0));
        else
            MacChain<CNT>::vv0(acc, x + I0, wq + (C - I0));
    } else if constexpr (UNIFORM)
        MacChain<CNT>::vs(acc, x + I0, wq + (C - I0));
    else
        MacChain<CNT>::vv(acc, x + I0, wq + (C - I0));
    if constexpr (C >= N) q[C - N] = (uint32_t)acc & ((1u << 29) - 1);
    acc >>= 29;
}
template <class F, bool UNIFORM, int... C>
__device__ inline __attribute__((always_inline)) void fe29_shoup_hi_cols(uint64_t &acc, const uint32_t *x, const uint32_t *wq, uint32_t *q, std::integer_sequence<int, C...>)
{
    (fe29_shoup_hi_col<F, UNIFORM, F::N - 2 + C>(acc, x, wq, q), ...);
}
template <class F, bool UNIFORM, int C>
__device__ inline __attribute__((always_inline)) void fe29_shoup_lo_col(uint64_t &acc, const uint32_t *x, const uint32_t *w, const uint32_t *q, uint32_t *out)
{
    if constexpr (C == 0) { // column 0 of the low halves starts the accumulator afresh
        if constexpr (UNIFORM)
            MacChain<1>::vs0(acc, x, w);
        else
            MacChain<1>::vv0(acc, x, w);
    } else if constexpr (UNIFORM)
        MacChain<C + 1>::vs(acc, x, w + C);
    else
        MacChain<C + 1>::vv(acc, x, w + C);
    MacChain<C + 1>::vs(acc, q, &F::PNEG[C]);
    out[C] = (uint32_t)acc & ((1u << 29) - 1);
    if constexpr (C + 1 < F::N) acc >>= 29;
}
template <class F, bool UNIFORM, int... C>
__device__ inline __attribute__((always_inline)) void fe29_shoup_lo_cols(uint64_t &acc, const uint32_t *x, const uint32_t *w, const uint32_t *q, uint32_t *out,
                                                                         std::integer_sequence<int, C...>)
{
    (fe29_shoup_lo_col<F, UNIFORM, C>(acc, x, w, q, out), ...);
}
template <class F, bool UNIFORM, int... C>
__device__ inline __attribute__((always_inline)) void fe29_shoup_hi_cols_x2(uint64_t &a0, uint64_t &a1, const uint32_t *x0, const uint32_t *x1, const uint32_t *wq0,
                                                                            const uint32_t *wq1, uint32_t *q0, uint32_t *q1, std::integer_sequence<int, C...>)
{
    ((fe29_shoup_hi_col<F, UNIFORM, F::N - 2 + C>(a0, x0, wq0, q0), fe29_shoup_hi_col<F, UNIFORM, F::N - 2 + C>(a1, x1, wq1, q1)), ...);
}
template <class F, bool UNIFORM, int... C>
__device__ inline __attribute__((always_inline)) void fe29_shoup_lo_cols_x2(uint64_t &a0, uint64_t &a1, const uint32_t *x0, const uint32_t *x1, const uint32_t *w0,
                                                                            const uint32_t *w1, const uint32_t *q0, const uint32_t *q1, uint32_t *o0, uint32_t *o1,
                                                                            std::integer_sequence<int, C...>)
{
    ((fe29_shoup_lo_col<F, UNIFORM, C>(a0, x0, w0, q0, o0), fe29_shoup_lo_col<F, UNIFORM, C>(a1, x1, w1, q1, o1)), ...);
}
#endif

// r = x * w - q * p.  UNIFORM: the constant is the same in every lane of the wave and sits in scalar registers.
template <class F, bool UNIFORM = false>
PANDA_HD void fe_mul_shoup(Fe<F> &r, const Fe<F> &x, const u32 *w, const u32 *wq)
{
    constexpr int N = F::N;
    static_assert(RawOperandOk<F>::value, "no column headroom for a raw operand in this field");
    u32 q[N], out[N];
    u64 acc = 0;
#if defined(FE29_DEVICE_CHAINS)
    fe29_shoup_hi_cols<F, UNIFORM>(acc, x.l, wq, q, std::make_integer_sequence<int, N + 1>()); // columns N-2 .. 2N-2
    q[N - 1] = (u32)acc & LIMB_MASK;
    acc = 0;
    fe29_shoup_lo_cols<F, UNIFORM>(acc, x.l, w, q, out, std::make_integer_sequence<int, N>());
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
    return;
#endif
    {
        FE29_SHADOW_DECL
#pragma unroll
        for (int c = N - 2; c <= 2 * N - 2; c++) {
#pragma unroll
            for (int i = (c < N ? 0 : c - N + 1); i <= (c < N ? c : N - 1); i++) {
#if defined(FE29_CHECK)
                assert(x.l[i] < 3 * (1ull << 30) + 64 && wq[c - i] < (1u << 29) && "fe_mul_shoup operand range");
#endif
                FE29_MAC(acc, x.l[i], wq[c - i]);
                FE29_SHADOW_MAC(x.l[i], wq[c - i])
            }
            if (c >= N) q[c - N] = (u32)acc & LIMB_MASK;
            acc >>= LIMB_BITS;
            FE29_SHADOW_SHIFT()
        }
        q[N - 1] = (u32)acc & LIMB_MASK;
    }
    {
        acc = 0;
        FE29_SHADOW_DECL
#pragma unroll
        for (int c = 0; c < N; c++) {
#pragma unroll
            for (int i = 0; i <= c; i++) {
                FE29_MAC(acc, x.l[i], w[c - i]);
                FE29_SHADOW_MAC(x.l[i], w[c - i])
                FE29_MAC_CONST(acc, q[i], F::PNEG[c - i]);
                FE29_SHADOW_MAC(q[i], F::PNEG[c - i])
            }
            out[c] = (u32)acc & LIMB_MASK;
            acc >>= LIMB_BITS;
            FE29_SHADOW_SHIFT()
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = out[i];
}

template <class F, bool UNIFORM = false>
PANDA_HD void fe_mul_shoup(Fe<F> &r, const Fe<F> &x, const FeTw<F> &t)
{
    fe_mul_shoup<F, UNIFORM>(r, x, t.w, t.q);
}

// Two independent products with their columns interleaved: r0 = x0 * (w0, wq0), r1 = x1 * (w1, wq1).  A column is a dependent chain
// of multiply-adds (one wave alone issues them at 40 % of the instruction's rate), and hipcc puts a wait state between an asm block
// and the next instruction that reads its result; alternating the columns of two products gives every chain an independent
// neighbour, which removes both.
template <class F, bool UNIFORM = false>
PANDA_HD void fe_mul_shoup_x2(Fe<F> &r0, Fe<F> &r1, const Fe<F> &x0, const Fe<F> &x1, const u32 *w0, const u32 *wq0, const u32 *w1, const u32 *wq1)
{
#if defined(FE29_DEVICE_CHAINS)
    constexpr int N = F::N;
    static_assert(RawOperandOk<F>::value, "no column headroom for a raw operand in this field");
    u32 q0[N], q1[N], o0[N], o1[N];
    u64 a0 = 0, a1 = 0;
    fe29_shoup_hi_cols_x2<F, UNIFORM>(a0, a1, x0.l, x1.l, wq0, wq1, q0, q1, std::make_integer_sequence<int, N + 1>());
    q0[N - 1] = (u32)a0 & LIMB_MASK;
    q1[N - 1] = (u32)a1 & LIMB_MASK;
    a0 = a1 = 0;
    fe29_shoup_lo_cols_x2<F, UNIFORM>(a0, a1, x0.l, x1.l, w0, w1, q0, q1, o0, o1, std::make_integer_sequence<int, N>());
#pragma unroll
    for (int i = 0; i < N; i++) {
        r0.l[i] = o0[i];
        r1.l[i] = o1[i];
    }
#else
    fe_mul_shoup<F, UNIFORM>(r0, x0, w0, wq0);
    fe_mul_shoup<F, UNIFORM>(r1, x1, w1, wq1);
#endif
}

// (w, floor(w R / p)) from w in internal form (w R mod p, tight, < 2p).  w R = k p + (w R mod p) with k the quotient wanted, so
// k = (w R mod p) * (-p^-1) mod R: one truncated product.  Table construction only.
template <class F>
PANDA_HD void fe_shoup_prepare(FeTw<F> &t, const Fe<F> &w_internal)
{
    constexpr int N = F::N;
    Fe<F> wm = w_internal, one, wp;
    fe_reduce_once(wm);
    fe_zero(one);
    one.l[0] = 1;
    fe_mul(wp, wm, one); // w R / R
    fe_reduce_once(wp);
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < N; c++) {
#pragma unroll
        for (int i = 0; i <= c; i++) acc += (u64)wm.l[i] * F::NINV[c - i];
        t.q[c] = (u32)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
    }
#pragma unroll
    for (int i = 0; i < N; i++) t.w[i] = wp.l[i];
}

// limbs < 2^32, value < 2^9 p  ->  tight, [0, 2p): the multiply-add spelling of fe_reduce_small_2p without the carry pass in front.
// The quotient is estimated from the top limb as it stands: it never overshoots (a >= top 2^(29(N-1)), p < (top(p)+1) 2^(29(N-1)))
// and what it leaves is below p + (2^9 + 9) 2^(29(N-1)) < 2p (pending carries from below are worth less than 9 units of the top
// limb).  a - q p is taken as the low N limbs of a + q (R - p), so carries, subtraction and normalisation are one chain of N
// multiply-adds.
template <class F>
PANDA_HD void fe_reduce_mad_2p(Fe<F> &a)
{
    constexpr int N = F::N;
    static_assert(F::P[N - 1] >= (1u << 16), "fe_reduce_mad: top limb of p too narrow for the quotient estimate");
    constexpr u64 C = (1ull << 52) / ((u64)F::P[N - 1] + 1);
    const u32 q = (u32)(((u64)a.l[N - 1] * C) >> 52);
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        acc += a.l[i];
        FE29_MAC_CONST(acc, q, F::PNEG[i]);
        a.l[i] = (u32)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
    }
}

// any value within the fe_mul input contract -> canonical [0, p), same residue
template <class F>
PANDA_HD void fe_canon(Fe<F> &r, const Fe<F> &a)
{
    Fe<F> one;
    fe_one(one);
    fe_mul(r, a, one); // a * R / R, now tight and < 2p
    fe_reduce_once(r);
}

// residue zero?  (one multiply; used only on rare paths)
template <class F>
PANDA_HD bool fe_is_zero_mod_p(const Fe<F> &a)
{
    Fe<F> t;
    fe_canon(t, a);
    return fe_all_zero(t);
}

// tight value in [0, 2p): is it 0 or p?
template <class F>
PANDA_HD bool fe_is_zero_2p(const Fe<F> &a)
{
    u32 z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        z |= a.l[i];
        e |= a.l[i] ^ F::P[i];
    }
    return z == 0 || e == 0;
}

// wire (x * 2^(32L) mod p, 32-bit limbs) -> internal (x * 2^(29N) mod p), tight, < 2p
template <class F>
PANDA_HD void fe_from_wire(Fe<F> &r, const u32 *w)
{
    Fe<F> t, k;
    fe_unpack(t, w);
    fe_const(k, F::K_IN);
    fe_mul(r, t, k);
}

// internal -> wire, canonical
template <class F>
PANDA_HD void fe_to_wire(u32 *w, const Fe<F> &a)
{
    Fe<F> t, k;
    fe_const(k, F::K_OUT);
    fe_mul(t, a, k);
    fe_reduce_once(t);
    fe_pack(w, t);
}

// wire Montgomery scalar -> canonical integer (32-bit limbs out)
template <class F>
PANDA_HD void fe_wire_to_canonical(u32 *out, const u32 *w)
{
    Fe<F> t, k, c;
    fe_unpack(t, w);
    fe_const(k, F::K_CANON);
    fe_mul(c, t, k);
    fe_reduce_once(c);
    fe_pack(out, c);
}

// canonical small integer -> internal form
template <class F>
PANDA_HD void fe_from_u32(Fe<F> &r, u32 v)
{
    Fe<F> t, k;
    fe_zero(t);
    t.l[0] = v & LIMB_MASK;
    t.l[1] = v >> LIMB_BITS;
    fe_const(k, F::K_TOINT);
    fe_mul(r, t, k);
}

template <class F>
PANDA_HD void fe_select(Fe<F> &r, bool c, const Fe<F> &a, const Fe<F> &b)
{
#pragma unroll
    for (int i = 0; i < F::N; i++) r.l[i] = c ? a.l[i] : b.l[i];
}

// a^e for a 64-bit exponent (host-side helpers, twiddle setup)
template <class F>
PANDA_HD void fe_pow_u64(Fe<F> &r, const Fe<F> &a, u64 e)
{
    Fe<F> acc, base = a;
    fe_one(acc);
    while (e) {
        if (e & 1) fe_mul(acc, acc, base);
        fe_sqr(base, base);
        e >>= 1;
    }
    r = acc;
}

// a^(p-2): inverse by Fermat (host side / rare device paths)
template <class F>
PANDA_HD void fe_inv(Fe<F> &r, const Fe<F> &a)
{
    // exponent p - 2 from the 32-bit limb table
    u32 e[F::L];
#pragma unroll
    for (int i = 0; i < F::L; i++) e[i] = F::PW[i];
    u32 borrow = 2;
#pragma unroll
    for (int i = 0; i < F::L; i++) {
        u32 t = e[i] - borrow;
        borrow = e[i] < borrow ? 1u : 0u;
        e[i] = t;
    }
    Fe<F> acc, base = a;
    fe_one(acc);
#pragma unroll
    for (int w = 0; w < F::L; w++) { // words statically indexed: no scratch on the device
        u32 ew = e[w];
        for (int b = 0; b < 32; b++) {
            if ((ew >> b) & 1) fe_mul(acc, acc, base);
            fe_sqr(base, base);
        }
    }
    r = acc;
}

} // namespace panda29

#include "fe29_ext2.h"
