// fe29_ext2.h -- the quadratic extension B[u] / (u^2 + 1) on fe29 elements: the coordinate field of G2 for curves whose base
// field has -1 as a quadratic non-residue (BN254, BLS12-381).  Included at the end of fe29.h.
//
// Ext2<B> is a field descriptor like Bn254Fq, so curve29.h's group law and the MSM kernels instantiate over it unchanged:
// Fe<Ext2<B>> is c0's limbs followed by c1's (2 N limbs, 2 L wire words: c0 || c1, the order arkworks serialises Fp2 in).
// Every arithmetic function of fe29.h that carries between limbs has an overload here that works on the halves; each half
// meets the contract of the base function it stands in for (products tight and below 2p, sums and differences loose), so
// the per-formula bounds of curve29.h hold component by component.  Products are schoolbook:
//   (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u
// four base products, each within the very bound curve29.h asserts for the product it stands in for (fusing a0 b1 + a1 b0
// under one reduction would need twice that), then a subtraction / addition and a multiply-free reduction back below 2p.
// a b + c d fuses the products that belong to DIFFERENT terms -- a_i b_j + c_i d_j -- which is exactly the sum curve29.h
// bounds for fe_mul_add.  No raw (un-normalised) operands.
#pragma once

namespace panda29 {

template <class B>
struct Ext2 {
    typedef B Base;
    static constexpr int N = 2 * B::N;
    static constexpr int L = 2 * B::L;
    static constexpr int BITS = B::BITS;
    static constexpr long long HEADROOM = B::HEADROOM;
};

template <class B>
struct SubMargin<Ext2<B>> : SubMargin<B> {
};
template <class B, int KB>
struct SubGrowth<Ext2<B>, KB> : SubGrowth<B, KB> {
};
template <class B>
struct RawOperandOk<Ext2<B>> {
    static constexpr bool value = false;
};

template <class B>
PANDA_HD void fe_one(Fe<Ext2<B>> &r)
{
    fe_one(ext_c0(r));
    fe_zero(ext_c1(r));
}

template <class B>
PANDA_HD void fe_norm(Fe<Ext2<B>> &r, const Fe<Ext2<B>> &t)
{
    fe_norm(ext_c0(r), ext_c0(t));
    fe_norm(ext_c1(r), ext_c1(t));
}

template <class B>
PANDA_HD void fe_carry(Fe<Ext2<B>> &a)
{
    fe_carry(ext_c0(a));
    fe_carry(ext_c1(a));
}

template <class B>
PANDA_HD void fe_reduce_once(Fe<Ext2<B>> &a)
{
    fe_reduce_once(ext_c0(a));
    fe_reduce_once(ext_c1(a));
}

template <class B>
PANDA_HD void fe_reduce_small_2p(Fe<Ext2<B>> &a)
{
    fe_reduce_small_2p(ext_c0(a));
    fe_reduce_small_2p(ext_c1(a));
}

// (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u; components tight, < 2p.  Every a_i b_j within the base contract of fe_mul.
template <class B>
PANDA_HD void fe_mul(Fe<Ext2<B>> &r, const Fe<Ext2<B>> &a, const Fe<Ext2<B>> &b)
{
    Fe<B> t0, t1, t2, t3, c0, c1;
    fe_mul(t0, ext_c0(a), ext_c0(b));
    fe_mul(t1, ext_c1(a), ext_c1(b));
    fe_mul(t2, ext_c0(a), ext_c1(b));
    fe_mul(t3, ext_c1(a), ext_c0(b));
    fe_sub<B, 2>(c0, t0, t1); // < (2 + 2 + M) p, loose
    fe_reduce_small_2p(c0);   // tight, < 2p
    fe_add_nr(c1, t2, t3);    // < 4p, limbs < 2^30
    fe_reduce_small_2p(c1);
    ext_c0(r) = c0;
    ext_c1(r) = c1;
}

// (a0^2 - a1^2) + 2 a0 a1 u
template <class B>
PANDA_HD void fe_sqr(Fe<Ext2<B>> &r, const Fe<Ext2<B>> &a)
{
    Fe<B> t0, t1, c0, c1, m;
    fe_sqr(t0, ext_c0(a));
    fe_sqr(t1, ext_c1(a));
    fe_mul(m, ext_c0(a), ext_c1(a));
    fe_sub<B, 2>(c0, t0, t1);
    fe_reduce_small_2p(c0);
    fe_add_nr(c1, m, m); // < 4p, limbs < 2^30
    fe_reduce_small_2p(c1);
    ext_c0(r) = c0;
    ext_c1(r) = c1;
}

// a b + c d.  Each fused pair a_i b_j + c_i d_j is the sum the base fe_mul_add is specified for (value < 0.9 R p is what the
// caller's bound on a b + c d means component by component).
template <class B>
PANDA_HD void fe_mul_add(Fe<Ext2<B>> &r, const Fe<Ext2<B>> &a, const Fe<Ext2<B>> &b, const Fe<Ext2<B>> &c, const Fe<Ext2<B>> &d)
{
    Fe<B> p0, p1, q0, q1, c0, c1;
    fe_mul_add(p0, ext_c0(a), ext_c0(b), ext_c0(c), ext_c0(d)); // a0 b0 + c0 d0
    fe_mul_add(p1, ext_c1(a), ext_c1(b), ext_c1(c), ext_c1(d)); // a1 b1 + c1 d1
    fe_mul_add(q0, ext_c0(a), ext_c1(b), ext_c0(c), ext_c1(d)); // a0 b1 + c0 d1
    fe_mul_add(q1, ext_c1(a), ext_c0(b), ext_c1(c), ext_c0(d)); // a1 b0 + c1 d0
    fe_sub<B, 2>(c0, p0, p1);
    fe_reduce_small_2p(c0);
    fe_add_nr(c1, q0, q1);
    fe_reduce_small_2p(c1);
    ext_c0(r) = c0;
    ext_c1(r) = c1;
}

// both components 0 or p (tight values below 2p)
template <class B>
PANDA_HD bool fe_is_zero_2p(const Fe<Ext2<B>> &a)
{
    return fe_is_zero_2p(ext_c0(a)) && fe_is_zero_2p(ext_c1(a));
}

template <class B>
PANDA_HD void fe_unpack(Fe<Ext2<B>> &r, const u32 *w)
{
    fe_unpack(ext_c0(r), w);
    fe_unpack(ext_c1(r), w + B::L);
}

template <class B>
PANDA_HD void fe_pack(u32 *w, const Fe<Ext2<B>> &a)
{
    fe_pack(w, ext_c0(a));
    fe_pack(w + B::L, ext_c1(a));
}

template <class B>
PANDA_HD void fe_from_wire(Fe<Ext2<B>> &r, const u32 *w)
{
    fe_from_wire(ext_c0(r), w);
    fe_from_wire(ext_c1(r), w + B::L);
}

template <class B>
PANDA_HD void fe_to_wire(u32 *w, const Fe<Ext2<B>> &a)
{
    fe_to_wire(w, ext_c0(a));
    fe_to_wire(w + B::L, ext_c1(a));
}

template <class B>
PANDA_HD void fe_from_u32(Fe<Ext2<B>> &r, u32 v)
{
    fe_from_u32(ext_c0(r), v);
    fe_zero(ext_c1(r));
}

// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2); 0 -> 0
template <class B>
PANDA_HD void fe_inv(Fe<Ext2<B>> &r, const Fe<Ext2<B>> &a)
{
    Fe<B> n, ni, c0, c1, t;
    fe_mul_add(n, ext_c0(a), ext_c0(a), ext_c1(a), ext_c1(a));
    fe_inv(ni, n);
    fe_mul(c0, ext_c0(a), ni);
    fe_mul(t, ext_c1(a), ni);
    fe_neg<B, 2>(c1, t);
    fe_reduce_small_2p(c1);
    ext_c0(r) = c0;
    ext_c1(r) = c1;
}

} // namespace panda29
