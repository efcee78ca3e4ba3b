// curve29.h -- short-Weierstrass a = 0 group law on fe29 elements (host + device).
//
// Replaces the reference's Jacobian formulas (src/cuda/core/curve/projective.cuh:163-314:
// dbl_2009_l 2M+5S, add_2007_bl 11M+5S, madd_2007_bl 7M+4S) with extended-Jacobian "XYZZ"
// coordinates (X, Y, ZZ, ZZZ; x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), which make the dominant
// operation of the bucket phase -- accumulator += affine base -- 8M+2S and need no field
// inversion or special-casing of Z.  Only affine values are part of the parity contract
// (tests/test.rs:101-108), so the internal representative is free.
//
// Exceptional cases are handled exactly, as in the reference (projective.cuh:203-231, 263-288):
//   identity operands, P + P -> doubling, P + (-P) -> identity.  The reference's own k13 fixture
//   (8192 x the generator) consists of nothing but the P + P case.
// Identity is encoded as ZZ == 0 (all limbs zero), mirroring "z == 0" (projective.cuh:111-114).
//
// Lazy-reduction invariants on stored points (units of p; M = SubMargin<F>):
//   X < (8+M) p loose, Y < YB p (2p after an addition; k p - y when the accumulator was seeded with a negated base), ZZ, ZZZ < 2p tight.
// Each formula below lists the bound of every intermediate; fe_mul needs value(a)*value(b) < 0.9 R p,
// i.e. (bound a)*(bound b) < 0.9 * HEADROOM, which the static_asserts check per field.
#pragma once
#include "fe29.h"

namespace panda29 {

template <class F>
struct Xyzz {
    Fe<F> X, Y, ZZ, ZZZ;
};

template <class F>
struct Bounds {
    static constexpr int M = SubMargin<F>::value;
    static constexpr int XB = 8 + M;  // stored X (loose)
    static constexpr int YB = SubGrowth<F, 2>::value; // stored Y: < 2p out of fe_mul_add, or a negated base coordinate k p - y
    static constexpr int PB = 2 + SubGrowth<F, XB>::value; // P = U2 - X1 + k p
    static constexpr int RB = 2 + SubGrowth<F, YB>::value; // R = S2 - Y1 + k p
    static constexpr int VB = 2 + SubGrowth<F, XB>::value; // Q - X3 + k p
    static constexpr long long LIM = F::HEADROOM * 9 / 10;
    static_assert((long long)PB * PB < LIM, "P^2 exceeds the lazy-reduction headroom");
    static constexpr int NB = SubGrowth<F, 2>::value; // -PPP, -W: k p - (a value below 2p)
    static_assert((long long)RB * VB + (long long)YB * NB < LIM, "R*(Q-X3) - Y1*PPP exceeds the lazy-reduction headroom");
    static_assert(6LL * VB + 2LL * NB < LIM, "3x^2*(S-X3) - W*Y exceeds the lazy-reduction headroom");
    static_assert((long long)(2 * YB) * (2 * YB) < LIM, "(2Y)^2 exceeds the lazy-reduction headroom");
    static_assert(2 + SubGrowth<F, 6>::value <= XB, "X3 bound");
};

template <class F>
PANDA_HD void xyzz_set_identity(Xyzz<F> &p)
{
    fe_zero(p.X);
    fe_zero(p.Y);
    fe_zero(p.ZZ);
    fe_zero(p.ZZZ);
}

template <class F>
PANDA_HD bool xyzz_is_identity(const Xyzz<F> &p)
{
    return fe_all_zero(p.ZZ);
}

template <class F>
PANDA_HD void xyzz_from_affine(Xyzz<F> &p, const Fe<F> &x, const Fe<F> &y)
{
    p.X = x;
    p.Y = y;
    fe_one(p.ZZ);
    fe_one(p.ZZZ);
}

// shared tail of add / madd: given P, R (loose), U1 (= X1 or X1*ZZ2), S1, PP, PPP computes X3, Y3
template <class F>
PANDA_HD void xyzz_finish(Fe<F> &X3, Fe<F> &Y3, const Fe<F> &R, const Fe<F> &Q, const Fe<F> &PPP, const Fe<F> &S1)
{
    typedef Bounds<F> B;
    Fe<F> t, rr, v, nppp;
    fe_add_nr(t, Q, Q);       // 2Q      < 4p raw
    fe_add_nr(t, t, PPP);     // + PPP   < 6p, limbs < 3*2^29
    fe_sqr(rr, R);            // R^2     < 2p
    fe_sub<F, 6>(X3, rr, t);  // X3      < (2 + 6+M) p = XB p
    fe_sub<F, B::XB>(v, Q, X3); // Q - X3 < VB p
    if constexpr (RawOperandOk<F>::value)
        fe_neg_raw<F, 2>(nppp, PPP); // k p - PPP < NB p, limbs < 2^31: it only feeds the product below
    else
        fe_neg<F, 2>(nppp, PPP);
    fe_mul_add(Y3, R, v, S1, nppp); // (R (Q - X3) - S1 PPP) / R: one reduction for both products; tight, < 2p
}

// 2 * (x, y) for an affine point (mdbl-2008-s-1); x, y tight < 2p
template <class F>
PANDA_HD void xyzz_dbl_affine(Xyzz<F> &r, const Fe<F> &x, const Fe<F> &y)
{
    typedef Bounds<F> B;
    Fe<F> U, V, W, S, A, M3, t, v, t1, t2;
    fe_add(U, y, y); // 2y < 4p loose (a raw operand may not be squared when N = 14)
    fe_sqr(V, U);
    if (fe_is_zero_2p(V)) { // y == 0: a point of order two
        xyzz_set_identity(r);
        return;
    }
    fe_mul(W, U, V);
    fe_mul(S, x, V);
    fe_sqr(A, x);
    fe_add_nr(t, A, A);
    fe_add_nr(t, t, A);
    fe_norm(M3, t); // 3x^2 < 6p loose
    fe_sqr(t1, M3);
    fe_add_nr(t, S, S);           // 2S < 4p raw
    fe_sub<F, 4>(r.X, t1, t);     // < (2+4+M) p <= XB p
    fe_sub<F, B::XB>(v, S, r.X);  // < VB p
    fe_neg<F, 2>(t2, W);
    fe_mul_add(r.Y, M3, v, t2, y); // (3x^2 (S - X3) - W y) / R
    r.ZZ = V;
    r.ZZZ = W;
}

// r = 2 p (dbl-2008-s-1)
template <class F>
PANDA_HD void xyzz_dbl(Xyzz<F> &r, const Xyzz<F> &p)
{
    typedef Bounds<F> B;
    if (xyzz_is_identity(p)) {
        xyzz_set_identity(r);
        return;
    }
    Fe<F> U, V, W, S, A, M3, t, v, t1, t2, X3, Y3;
    fe_add(U, p.Y, p.Y); // < 2 YB p loose
    fe_sqr(V, U);
    if (fe_is_zero_2p(V)) {
        xyzz_set_identity(r);
        return;
    }
    fe_mul(W, U, V);
    fe_mul(S, p.X, V);
    fe_sqr(A, p.X); // XB^2 < LIM since XB < PB
    fe_add_nr(t, A, A);
    fe_add_nr(t, t, A);
    fe_norm(M3, t);
    fe_sqr(t1, M3);
    fe_add_nr(t, S, S);
    fe_sub<F, 4>(X3, t1, t);
    fe_sub<F, B::XB>(v, S, X3);
    fe_neg<F, 2>(t2, W);
    fe_mul_add(Y3, M3, v, t2, p.Y);
    fe_mul(t, V, p.ZZ);
    fe_mul(t2, W, p.ZZZ);
    r.X = X3;
    r.Y = Y3;
    r.ZZ = t;
    r.ZZZ = t2;
}

// acc += (bx, by) for a non-identity accumulator and base; bx tight < 2p, by tight or (RawOperandOk fields) the raw31
// output of fe_neg_raw: it only feeds the product by * ZZZ.  8M + 2S.
// Returns 0 when done.  When the two points share their x coordinate nothing is written and the caller finishes the
// job: 1 = same point (double the BASE, xyzz_dbl_affine), 2 = opposite points (the sum is the identity).
// Splitting the rare cases off keeps bx, by dead after the first two products (register pressure in k_accumulate).
template <class F>
PANDA_HD int xyzz_madd_core(Xyzz<F> &acc, const Fe<F> &bx, const Fe<F> &by)
{
    typedef Bounds<F> B;
    Fe<F> U2, S2, P, R, PP, PPP, Q, X3, Y3;
    fe_mul(U2, bx, acc.ZZ);
    fe_sub<F, B::XB>(P, U2, acc.X); // < PB p
    fe_mul(S2, by, acc.ZZZ);
    fe_sub<F, B::YB>(R, S2, acc.Y); // < RB p
    fe_sqr(PP, P);
    if (fe_is_zero_2p(PP)) // same x: P + P or P + (-P)   (projective.cuh:284-288)
        return fe_is_zero_mod_p(R) ? 1 : 2;
    fe_mul(PPP, P, PP);
    fe_mul(Q, acc.X, PP);
    fe_mul(acc.ZZ, acc.ZZ, PP);
    fe_mul(acc.ZZZ, acc.ZZZ, PPP);
    xyzz_finish(X3, Y3, R, Q, PPP, acc.Y);
    acc.X = X3;
    acc.Y = Y3;
    return 0;
}

// acc += (bx, by); base_identity <=> the wire x was 0 (affine.cuh:72-75)
template <class F>
PANDA_HD void xyzz_madd(Xyzz<F> &acc, const Fe<F> &bx, const Fe<F> &by, bool base_identity)
{
    if (base_identity) return;
    if (xyzz_is_identity(acc)) {
        xyzz_from_affine(acc, bx, by);
        return;
    }
    const int rare = xyzz_madd_core(acc, bx, by);
    if (rare == 1)
        xyzz_dbl_affine(acc, bx, by);
    else if (rare == 2)
        xyzz_set_identity(acc);
}

// acc += q  (add-2008-s, 12M + 2S)
template <class F>
PANDA_HD void xyzz_add(Xyzz<F> &acc, const Xyzz<F> &q)
{
    if (xyzz_is_identity(q)) return;
    if (xyzz_is_identity(acc)) {
        acc = q;
        return;
    }
    Fe<F> U1, U2, S1, S2, P, R, PP, PPP, Q, X3, Y3, t;
    fe_mul(U1, acc.X, q.ZZ);
    fe_mul(U2, q.X, acc.ZZ);
    fe_mul(S1, acc.Y, q.ZZZ);
    fe_mul(S2, q.Y, acc.ZZZ);
    fe_sub<F, 2>(P, U2, U1);
    fe_sub<F, 2>(R, S2, S1);
    fe_sqr(PP, P);
    if (fe_is_zero_2p(PP)) {
        if (fe_is_zero_mod_p(R)) {
            Xyzz<F> d;
            xyzz_dbl(d, acc);
            acc = d;
        } else
            xyzz_set_identity(acc);
        return;
    }
    fe_mul(PPP, P, PP);
    fe_mul(Q, U1, PP);
    xyzz_finish(X3, Y3, R, Q, PPP, S1);
    fe_mul(t, acc.ZZ, q.ZZ);
    fe_mul(acc.ZZ, t, PP);
    fe_mul(t, acc.ZZZ, q.ZZZ);
    fe_mul(acc.ZZZ, t, PPP);
    acc.X = X3;
    acc.Y = Y3;
}

// acc -= (bx, by) is madd with the base's y negated: y' = k p - y, loose
template <class F>
PANDA_HD void fe_neg_tight2p(Fe<F> &r, const Fe<F> &a)
{
    fe_neg<F, 2>(r, a);
}

// XYZZ -> Jacobian triple with Z = ZZ: (X*ZZ, Y*ZZZ, ZZ); x = X'/Z'^2, y = Y'/Z'^3.  Wire limbs out.
template <class F>
PANDA_HD void xyzz_to_jacobian_wire(u32 *out, const Xyzz<F> &p)
{
    constexpr int L = F::L;
    if (xyzz_is_identity(p)) {
        // all zero, as the reference host path leaves it (`Projective h_result = {0x0}` sums, msm_host.cuh:218-234); only Z == 0 matters
#pragma unroll
        for (int i = 0; i < 3 * L; i++) out[i] = 0;
        return;
    }
    Fe<F> x, y;
    fe_mul(x, p.X, p.ZZ);
    fe_mul(y, p.Y, p.ZZZ);
    fe_to_wire(out, x);
    fe_to_wire(out + L, y);
    fe_to_wire(out + 2 * L, p.ZZ);
}

// XYZZ -> homogeneous projective (X*ZZZ, Y*ZZ, ZZ*ZZZ); x = X'/Z', y = Y'/Z'
// (what Projective::to_projective produces from a Jacobian point, projective.cuh:66-77, up to scaling)
template <class F>
PANDA_HD void xyzz_to_homogeneous_wire(u32 *out, const Xyzz<F> &p)
{
    constexpr int L = F::L;
    if (xyzz_is_identity(p)) {
#pragma unroll
        for (int i = 0; i < 3 * L; i++) out[i] = 0;
        return;
    }
    Fe<F> x, y, z;
    fe_mul(x, p.X, p.ZZZ);
    fe_mul(y, p.Y, p.ZZ);
    fe_mul(z, p.ZZ, p.ZZZ);
    fe_to_wire(out, x);
    fe_to_wire(out + L, y);
    fe_to_wire(out + 2 * L, z);
}

// non-identity XYZZ -> affine in internal form (tight, < 2p): one inversion, 1/ZZ = (ZZ/ZZZ)^2
template <class F>
PANDA_HD void xyzz_to_affine_internal(Fe<F> &x, Fe<F> &y, const Xyzz<F> &p)
{
    Fe<F> zi3, t, zi2;
    fe_inv(zi3, p.ZZZ);
    fe_mul(t, p.ZZ, zi3);
    fe_sqr(zi2, t);
    fe_mul(x, p.X, zi2);
    fe_mul(y, p.Y, zi3);
}

// Jacobian wire triple -> XYZZ (X, Y, Z^2, Z^3)
template <class F>
PANDA_HD void xyzz_from_jacobian_wire(Xyzz<F> &p, const u32 *in)
{
    constexpr int L = F::L;
    Fe<F> z;
    fe_from_wire(z, in + 2 * L);
    if (fe_is_zero_2p(z)) {
        xyzz_set_identity(p);
        return;
    }
    fe_from_wire(p.X, in);
    fe_from_wire(p.Y, in + L);
    fe_sqr(p.ZZ, z);
    fe_mul(p.ZZZ, p.ZZ, z);
}

// affine wire point (x||y, identity <=> x == 0) -> internal coordinates
template <class F>
PANDA_HD bool affine_from_wire(Fe<F> &x, Fe<F> &y, const u32 *in)
{
    constexpr int L = F::L;
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < L; i++) nz |= in[i];
    fe_from_wire(x, in);
    fe_from_wire(y, in + L);
    return nz == 0;
}

} // namespace panda29
