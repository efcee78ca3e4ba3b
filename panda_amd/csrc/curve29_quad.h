// curve29_quad.h -- XYZZ addition and doubling by FOUR cooperating lanes (device only).
//
// curve29.h's xyzz_add is 12M + 2S in sequence: 7.3 us for a wave that runs alone on its SIMD (it is issue-bound even so), which is what every level of
// the trees behind k_accumulate (fix-up, bucket reduction) costs -- some thirty levels per call, whatever the input size.  The
// formulas are only four products deep, so a quad of adjacent lanes takes one product each per step:
//
//   step 1   U1 = X1 ZZ2     U2 = X2 ZZ1     S1 = Y1 ZZZ2     S2 = Y2 ZZZ1          P = U2 - U1,  R = S2 - S1
//   step 2   PP = P P        RR = R R        zz = ZZ1 ZZ2     zzz = ZZZ1 ZZZ2       (P = 0: the exceptional cases, as xyzz_add)
//   step 3   PPP = P PP      Q = U1 PP       ZZ3 = zz PP      --                    X3 = RR - PPP - 2Q
//   step 4   Y3 = R (Q - X3) - S1 PPP        ZZZ3 = zzz PPP   (one fe_mul_add: lane 1's second product is 0 * 0)
//
// All four lanes hold the SAME operands and end with the same sum ("replicated in, replicated out"): a lane picks its factors with
// selects on (lane & 3), multiplies, and the four products travel to every lane of the quad as DPP quad_perm broadcasts -- no LDS, no
// barrier, nothing a divergent neighbour quad could disturb.  Bounds are those of xyzz_add / xyzz_dbl line by line (same fe_sub
// parameters, squarings taken as products of equal factors, which the squaring contract covers).
// Control flow must be uniform within a quad (it is: every decision is taken on replicated values).
#pragma once
#include <hip/hip_runtime.h>

#include "curve29.h"

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

namespace panda29 {

template <int K>
__device__ __forceinline__ u32 quad_bcast_u32(u32 v)
{
    // quad_perm [K, K, K, K]; all rows and banks enabled
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, false);
}

template <int K, class F>
__device__ __forceinline__ void quad_bcast(Fe<F> &r, const Fe<F> &v)
{
#pragma unroll
    for (int i = 0; i < F::N; i++) r.l[i] = quad_bcast_u32<K>(v.l[i]);
}

// role 0..3 -> a0..a3
template <class F>
__device__ __forceinline__ void quad_pick(Fe<F> &r, unsigned role, const Fe<F> &a0, const Fe<F> &a1, const Fe<F> &a2, const Fe<F> &a3)
{
    const bool odd = role & 1u, high = role & 2u;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        const u32 lo = odd ? a1.l[i] : a0.l[i];
        const u32 hi = odd ? a3.l[i] : a2.l[i];
        r.l[i] = high ? hi : lo;
    }
}

// acc += q; acc, q and the result replicated over the quad; role = lane & 3
template <class F>
__device__ __forceinline__ void xyzz_add_quad(Xyzz<F> &acc, const Xyzz<F> &q, unsigned role)
{
    typedef Bounds<F> B;
    if (xyzz_is_identity(q)) return;
    if (xyzz_is_identity(acc)) {
        acc = q;
        return;
    }
    Fe<F> a, b, r, U1, U2, S1, S2, P, R, PP, RR, zz, zzz, PPP, Q, t, X3, v, nppp, zero;
    // step 1
    quad_pick(a, role, acc.X, q.X, acc.Y, q.Y);
    quad_pick(b, role, q.ZZ, acc.ZZ, q.ZZZ, acc.ZZZ);
    fe_mul(r, a, b);
    quad_bcast<0>(U1, r);
    quad_bcast<1>(U2, r);
    quad_bcast<2>(S1, r);
    quad_bcast<3>(S2, r);
    fe_sub<F, 2>(P, U2, U1);
    fe_sub<F, 2>(R, S2, S1);
    // step 2
    quad_pick(a, role, P, R, acc.ZZ, acc.ZZZ);
    quad_pick(b, role, P, R, q.ZZ, q.ZZZ);
    fe_mul(r, a, b);
    quad_bcast<0>(PP, r);
    if (fe_is_zero_2p(PP)) { // same x: P + P or P + (-P), every lane on its own copy (rare)
        if (fe_is_zero_mod_p(R)) {
            Xyzz<F> d;
            xyzz_dbl(d, acc);
            acc = d;
        } else
            xyzz_set_identity(acc);
        return;
    }
    quad_bcast<1>(RR, r);
    quad_bcast<2>(zz, r);
    quad_bcast<3>(zzz, r);
    // step 3
    quad_pick(a, role, P, U1, zz, zz);
    fe_mul(r, a, PP);
    quad_bcast<0>(PPP, r);
    quad_bcast<1>(Q, r);
    quad_bcast<2>(acc.ZZ, r);
    fe_add_nr(t, Q, Q);     // 2Q      < 4p raw
    fe_add_nr(t, t, PPP);   // + PPP   < 6p
    fe_sub<F, 6>(X3, RR, t);
    fe_sub<F, B::XB>(v, Q, X3);
    if constexpr (RawOperandOk<F>::value)
        fe_neg_raw<F, 2>(nppp, PPP);
    else
        fe_neg<F, 2>(nppp, PPP);
    // step 4: lane 0  R v + S1 (-PPP);  lane 1  zzz PPP + 0 * 0.  The raw operand stays the LAST one (fe_sub_raw's contract).
    fe_zero(zero);
    const bool first = role == 0;
    fe_select(a, first, R, zzz);
    fe_select(b, first, v, PPP);
    fe_select(t, first, S1, zero);
    fe_select(r, first, nppp, zero);
    Fe<F> y;
    fe_mul_add(y, a, b, t, r);
    acc.X = X3;
    quad_bcast<0>(acc.Y, y);
    quad_bcast<1>(acc.ZZZ, y);
}

// r = 2 p, replicated over the quad (dbl-2008-s-1 as xyzz_dbl):
//   step 1   V = U U (U = 2Y)     A = X X                                 M = 3A
//   step 2   W = U V              S = X V          MM = M M      ZZ3 = V ZZ      X3 = MM - 2S
//   step 3   Y3 = M (S - X3) - W Y                 ZZZ3 = W ZZZ
template <class F>
__device__ __forceinline__ void xyzz_dbl_quad(Xyzz<F> &r, const Xyzz<F> &p, unsigned role)
{
    typedef Bounds<F> B;
    if (xyzz_is_identity(p)) {
        xyzz_set_identity(r);
        return;
    }
    Fe<F> U, a, b, m, V, A, M3, t, W, S, MM, X3, v, nW, zero, c, d, y;
    fe_add(U, p.Y, p.Y);
    // step 1
    fe_select(a, (role & 1u) != 0, p.X, U);
    fe_mul(m, a, a);
    quad_bcast<0>(V, m);
    if (fe_is_zero_2p(V)) { // y == 0: a point of order two
        xyzz_set_identity(r);
        return;
    }
    quad_bcast<1>(A, m);
    fe_add_nr(t, A, A);
    fe_add_nr(t, t, A);
    fe_norm(M3, t);
    // step 2
    quad_pick(a, role, U, p.X, M3, V);
    quad_pick(b, role, V, V, M3, p.ZZ);
    fe_mul(m, a, b);
    quad_bcast<0>(W, m);
    quad_bcast<1>(S, m);
    quad_bcast<2>(MM, m);
    quad_bcast<3>(r.ZZ, m);
    fe_add_nr(t, S, S);
    fe_sub<F, 4>(X3, MM, t);
    fe_sub<F, B::XB>(v, S, X3);
    fe_neg<F, 2>(nW, W);
    // step 3
    fe_zero(zero);
    const bool first = role == 0;
    fe_select(a, first, M3, W);
    fe_select(b, first, v, p.ZZZ);
    fe_select(c, first, nW, zero);
    fe_select(d, first, p.Y, zero);
    fe_mul_add(y, a, b, c, d);
    r.X = X3;
    quad_bcast<0>(r.Y, y);
    quad_bcast<1>(r.ZZZ, y);
}

} // namespace panda29

#endif
