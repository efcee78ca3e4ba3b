// Host build of the product's fe29.h / curve29.h with FE29_CHECK instrumentation (128-bit shadow
// accumulators, limb-range asserts).  Test infrastructure: compiled by tests/test_fe29_host.py with g++.
#define FE29_CHECK 1
#include "../../panda_amd/csrc/curve29.h"

#include <stddef.h>
#include <string.h>

using namespace panda29;

template <class F>
static void field_op(int op, u32 *r, const u32 *a, const u32 *b, size_t n)
{
    constexpr int L = F::L;
    for (size_t i = 0; i < n; i++) {
        Fe<F> x, y, z;
        if (op == 4) { // to_montgomery: canonical in -> wire Montgomery out
            Fe<F> t, k;
            fe_unpack(t, a + i * L);
            fe_const(k, F::K_TOINT);
            fe_mul(x, t, k); // internal form of the canonical integer
            fe_to_wire(r + i * L, x);
            continue;
        }
        if (op == 5) { // from_montgomery
            fe_wire_to_canonical<F>(r + i * L, a + i * L);
            continue;
        }
        fe_from_wire(x, a + i * L);
        if (b) fe_from_wire(y, b + i * L);
        switch (op) {
        case 0: fe_add(z, x, y); break;
        case 1: fe_sub<F, 2>(z, x, y); break;
        case 2: fe_mul(z, x, y); break;
        case 3: fe_sqr(z, x); break;
        case 6: fe_inv(z, x); break;
        }
        fe_to_wire(r + i * L, z);
    }
}

// Jacobian wire in/out; b is affine wire for op 0
template <class F>
static void curve_op(int op, u32 *r, const u32 *a, const u32 *b, size_t n)
{
    constexpr int L = F::L;
    for (size_t i = 0; i < n; i++) {
        Xyzz<F> p, q;
        xyzz_from_jacobian_wire(p, a + i * 3 * L);
        if (op == 0) {
            Fe<F> x, y;
            bool inf = affine_from_wire(x, y, b + i * 2 * L);
            xyzz_madd(p, x, y, inf);
        } else if (op == 1) {
            xyzz_from_jacobian_wire(q, b + i * 3 * L);
            xyzz_add(p, q);
        } else {
            xyzz_dbl(q, p);
            p = q;
        }
        xyzz_to_jacobian_wire(r + i * 3 * L, p);
    }
}

// acc = sum_i (+/-) base_i, one long dependent chain (stresses the lazy bounds); signs bit i of `neg`
template <class F>
static void chain(u32 *r, const u32 *bases, const unsigned char *neg, size_t n, int homogeneous)
{
    constexpr int L = F::L;
    Xyzz<F> acc;
    xyzz_set_identity(acc);
    for (size_t i = 0; i < n; i++) {
        Fe<F> x, y, ny;
        bool inf = affine_from_wire(x, y, bases + i * 2 * L);
        if (neg && neg[i]) {
            fe_neg_tight2p(ny, y);
            y = ny;
        }
        xyzz_madd(acc, x, y, inf);
    }
    if (homogeneous)
        xyzz_to_homogeneous_wire(r, acc);
    else
        xyzz_to_jacobian_wire(r, acc);
}

// raw 29-bit-limb interface to the precomputed-quotient product: x any limbs within the contract, w in internal form
template <class F>
static void shoup(u32 *r, u32 *tw_out, const u32 *x, const u32 *w_internal, size_t n)
{
    constexpr int N = F::N;
    for (size_t i = 0; i < n; i++) {
        Fe<F> xe, we, re;
        for (int j = 0; j < N; j++) {
            xe.l[j] = x[i * N + j];
            we.l[j] = w_internal[i * N + j];
        }
        FeTw<F> t;
        fe_shoup_prepare(t, we);
        fe_mul_shoup(re, xe, t);
        for (int j = 0; j < N; j++) {
            r[i * N + j] = re.l[j];
            tw_out[i * 2 * N + j] = t.w[j];
            tw_out[i * 2 * N + N + j] = t.q[j];
        }
    }
}
// the NTT butterfly's difference path: (a - b + K p) with the U * 2^29 bias, un-normalised, straight into the product
template <class F, int U>
static void bfly_diff(u32 *r, const u32 *a, const u32 *b, const u32 *w_internal, size_t n)
{
    constexpr int N = F::N;
    for (size_t i = 0; i < n; i++) {
        Fe<F> ae, be, we, x, re;
        for (int j = 0; j < N; j++) {
            ae.l[j] = a[i * N + j];
            be.l[j] = b[i * N + j];
            we.l[j] = w_internal[i * N + j];
        }
        FeTw<F> t;
        fe_shoup_prepare(t, we);
        fe_sub_raw_bias<F, 8, U>(x, ae, be);
        fe_mul_shoup(re, x, t);
        for (int j = 0; j < N; j++) r[i * N + j] = re.l[j];
    }
}

template <class F>
static void reduce_mad(u32 *r, const u32 *x, size_t n)
{
    constexpr int N = F::N;
    for (size_t i = 0; i < n; i++) {
        Fe<F> xe;
        for (int j = 0; j < N; j++) xe.l[j] = x[i * N + j];
        fe_reduce_mad_2p(xe);
        for (int j = 0; j < N; j++) r[i * N + j] = xe.l[j];
    }
}

extern "C" {
int h29_shoup(int field_id, u32 *r, u32 *tw_out, const u32 *x, const u32 *w_internal, size_t n)
{
    switch (field_id) {
    case 1: shoup<Bn254Fr>(r, tw_out, x, w_internal, n); return 0;
    case 3: shoup<Bls377Fr>(r, tw_out, x, w_internal, n); return 0;
    case 5: shoup<Bls381Fr>(r, tw_out, x, w_internal, n); return 0;
    }
    return 1;
}
int h29_bfly_diff(int field_id, int bias_units, u32 *r, const u32 *a, const u32 *b, const u32 *w_internal, size_t n)
{
    if (bias_units != 2 && bias_units != 3) return 1;
    switch (field_id) {
    case 1: bias_units == 2 ? bfly_diff<Bn254Fr, 2>(r, a, b, w_internal, n) : bfly_diff<Bn254Fr, 3>(r, a, b, w_internal, n); return 0;
    case 3: bias_units == 2 ? bfly_diff<Bls377Fr, 2>(r, a, b, w_internal, n) : bfly_diff<Bls377Fr, 3>(r, a, b, w_internal, n); return 0;
    case 5: bias_units == 2 ? bfly_diff<Bls381Fr, 2>(r, a, b, w_internal, n) : bfly_diff<Bls381Fr, 3>(r, a, b, w_internal, n); return 0;
    }
    return 1;
}
int h29_reduce_mad(int field_id, u32 *r, const u32 *x, size_t n)
{
    switch (field_id) {
    case 1: reduce_mad<Bn254Fr>(r, x, n); return 0;
    case 3: reduce_mad<Bls377Fr>(r, x, n); return 0;
    case 5: reduce_mad<Bls381Fr>(r, x, n); return 0;
    }
    return 1;
}
int h29_field_op(int field_id, int op, u32 *r, const u32 *a, const u32 *b, size_t n)
{
    switch (field_id) {
    case 0: field_op<Bn254Fq>(op, r, a, b, n); return 0;
    case 1: field_op<Bn254Fr>(op, r, a, b, n); return 0;
    case 2: field_op<Bls377Fq>(op, r, a, b, n); return 0;
    case 3: field_op<Bls377Fr>(op, r, a, b, n); return 0;
    }
    return 1;
}
int h29_curve_op(int curve, int op, u32 *r, const u32 *a, const u32 *b, size_t n)
{
    if (curve == 0) curve_op<Bn254Fq>(op, r, a, b, n);
    else curve_op<Bls377Fq>(op, r, a, b, n);
    return 0;
}
int h29_chain(int curve, u32 *r, const u32 *bases, const unsigned char *neg, size_t n, int homogeneous)
{
    if (curve == 0) chain<Bn254Fq>(r, bases, neg, n, homogeneous);
    else chain<Bls377Fq>(r, bases, neg, n, homogeneous);
    return 0;
}
}
