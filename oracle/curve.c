/*
 * curve.c -- CPU ORACLE (test infrastructure, see panda_oracle.h): short-Weierstrass a=0
 * group law in Jacobian coordinates, restating src/cuda/core/curve/projective.cuh
 * (the class is *named* Projective but holds Jacobian X,Y,Z):
 *
 *   po_dbl            dbl_2009_l    projective.cuh:163-197
 *   po_add            add_2007_bl   projective.cuh:200-256
 *   po_madd           madd_2007_bl  projective.cuh:259-314
 *   po_to_affine      to_affine     projective.cuh:79-109   identity -> (0, R)
 *   po_to_projective  to_projective projective.cuh:66-77    (X*Z, Y, Z^3)
 *
 * The operation order inside each formula follows the reference line by line so that the
 * Jacobian triple (not only the affine point) equals the reference host path's.
 * Identity: Jacobian z == 0 (projective.cuh:111-114); affine x == 0 (affine.cuh:72-75).
 */
#include "panda_oracle.h"

#include <string.h>

typedef uint32_t u32;
typedef uint64_t u64;

#define LCMAX PO_MAX_LC

/* BLS12-377 G1 generator (arkworks ark-bls12-377 g1 generator), canonical form, little-endian limbs */
static const u32 BLS377_GX[12] = {0xb21be9efu, 0xeab9b16eu, 0xffcd394eu, 0xd5481512u, 0xbd37cb5cu, 0x188282c8u,
                                  0xaa9d41bbu, 0x85951e2cu, 0xbf87ff54u, 0xc8fc6225u, 0xfe740a67u, 0x008848deu};
static const u32 BLS377_GY[12] = {0x559c8ea6u, 0xfd82de55u, 0x34a9591au, 0xc2fe3d36u, 0x4fb82305u, 0x6d182ad4u,
                                  0xca3e52d9u, 0xbd7fb348u, 0x30afeec4u, 0x1f674f5du, 0xc5102effu, 0x01914a69u};

/* BLS12-381 G1 generator (the standard one), canonical form, little-endian limbs */
static const u32 BLS381_GX[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu, 0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
static const u32 BLS381_GY[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu, 0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};

void po_generator(int curve, u32 *aff)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    if (curve == PO_CURVE_BN254) {
        u32 one[LCMAX] = {1}, two[LCMAX] = {2};
        po_f_to_mont(f, aff, one);
        po_f_to_mont(f, aff + lc, two);
    } else if (curve == PO_CURVE_BLS12_381) {
        po_f_to_mont(f, aff, BLS381_GX);
        po_f_to_mont(f, aff + lc, BLS381_GY);
    } else {
        po_f_to_mont(f, aff, BLS377_GX);
        po_f_to_mont(f, aff + lc, BLS377_GY);
    }
}

int po_is_on_curve(int curve, const u32 *aff)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    u32 b[LCMAX] = {0}, bm[LCMAX], y2[LCMAX], x3[LCMAX];
    b[0] = (curve == PO_CURVE_BN254) ? 3 : (curve == PO_CURVE_BLS12_381 ? 4 : 1); /* y^2 = x^3 + b */
    po_f_to_mont(f, bm, b);
    po_f_sqr(f, y2, aff + lc);
    po_f_sqr(f, x3, aff);
    po_f_mul(f, x3, x3, aff);
    po_f_add(f, x3, x3, bm);
    return po_f_eq(f, y2, x3);
}

void po_dbl(int curve, u32 *r, const u32 *p)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    const u32 *px = p, *py = p + lc, *pz = p + 2 * lc;
    u32 rx[LCMAX], ry[LCMAX], rz[LCMAX], a[LCMAX], b[LCMAX], c[LCMAX], d[LCMAX], e[LCMAX], ff[LCMAX], c3[LCMAX];
    /* identity: z = 0 propagates through res.z = 2*y*z (projective.cuh:167-173) */
    po_f_mul(f, rz, py, pz);
    po_f_add(f, rz, rz, rz);
    po_f_sqr(f, a, px);
    po_f_sqr(f, b, py);
    po_f_sqr(f, c, b);
    po_f_add(f, d, px, b);
    po_f_sqr(f, d, d);
    po_f_sub(f, d, d, a);
    po_f_sub(f, d, d, c);
    po_f_add(f, d, d, d);
    po_f_add(f, e, a, a);
    po_f_add(f, e, e, a);
    po_f_sqr(f, ff, e);
    po_f_add(f, rx, d, d);
    po_f_sub(f, rx, ff, rx);
    po_f_sub(f, ry, d, rx);
    po_f_mul(f, ry, ry, e);
    po_f_add(f, c3, c, c);
    po_f_add(f, c3, c3, c3);
    po_f_add(f, c3, c3, c3);
    po_f_sub(f, ry, ry, c3);
    memcpy(r, rx, lc * 4);
    memcpy(r + lc, ry, lc * 4);
    memcpy(r + 2 * lc, rz, lc * 4);
}

void po_add(int curve, u32 *r, const u32 *p1, const u32 *p2)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    const u32 *x1 = p1, *y1 = p1 + lc, *z1 = p1 + 2 * lc;
    const u32 *x2 = p2, *y2 = p2 + lc, *z2 = p2 + 2 * lc;
    if (po_f_is_zero(f, z2)) {
        memmove(r, p1, 3 * lc * 4);
        return;
    }
    if (po_f_is_zero(f, z1)) {
        memmove(r, p2, 3 * lc * 4);
        return;
    }
    u32 z1z1[LCMAX], z2z2[LCMAX], u1[LCMAX], u2[LCMAX], s1[LCMAX], s2[LCMAX];
    po_f_sqr(f, z1z1, z1);
    po_f_sqr(f, z2z2, z2);
    po_f_mul(f, u1, x1, z2z2);
    po_f_mul(f, u2, x2, z1z1);
    po_f_mul(f, s1, y1, z2);
    po_f_mul(f, s1, s1, z2z2);
    po_f_mul(f, s2, y2, z1);
    po_f_mul(f, s2, s2, z1z1);
    if (po_f_eq(f, u1, u2) && po_f_eq(f, s1, s2)) {
        po_dbl(curve, r, p1);
        return;
    }
    u32 h[LCMAX], hh[LCMAX], i[LCMAX], j[LCMAX], rr[LCMAX], v[LCMAX], rx[LCMAX], ry[LCMAX], rz[LCMAX];
    po_f_sub(f, h, u2, u1);
    po_f_sqr(f, hh, h);
    po_f_add(f, i, hh, hh);
    po_f_add(f, i, i, i);
    po_f_mul(f, j, h, i);
    po_f_sub(f, rr, s2, s1);
    po_f_add(f, rr, rr, rr);
    po_f_mul(f, v, u1, i);
    po_f_sqr(f, rx, rr);
    po_f_sub(f, rx, rx, j);
    po_f_sub(f, rx, rx, v);
    po_f_sub(f, rx, rx, v);
    po_f_mul(f, j, s1, j);
    po_f_add(f, j, j, j);
    po_f_sub(f, ry, v, rx);
    po_f_mul(f, ry, ry, rr);
    po_f_sub(f, ry, ry, j);
    po_f_add(f, rz, z1, z2);
    po_f_sqr(f, rz, rz);
    po_f_sub(f, rz, rz, z1z1);
    po_f_sub(f, rz, rz, z2z2);
    po_f_mul(f, rz, rz, h);
    memcpy(r, rx, lc * 4);
    memcpy(r + lc, ry, lc * 4);
    memcpy(r + 2 * lc, rz, lc * 4);
}

void po_madd(int curve, u32 *r, const u32 *p1, const u32 *q)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    const u32 *x1 = p1, *y1 = p1 + lc, *z1 = p1 + 2 * lc;
    const u32 *x2 = q, *y2 = q + lc;
    if (po_f_is_zero(f, x2)) {
        memmove(r, p1, 3 * lc * 4);
        return;
    }
    if (po_f_is_zero(f, z1)) {
        u32 t[2 * LCMAX];
        memcpy(t, q, 2 * lc * 4);
        memcpy(r, t, 2 * lc * 4);
        memcpy(r + 2 * lc, f->one, lc * 4);
        return;
    }
    u32 z1z1[LCMAX], u2[LCMAX], s2[LCMAX];
    po_f_sqr(f, z1z1, z1);
    po_f_mul(f, u2, x2, z1z1);
    po_f_mul(f, s2, y2, z1);
    po_f_mul(f, s2, s2, z1z1);
    if (po_f_eq(f, x1, u2) && po_f_eq(f, y1, s2)) {
        po_dbl(curve, r, p1);
        return;
    }
    u32 h[LCMAX], hh[LCMAX], i[LCMAX], j[LCMAX], rr[LCMAX], v[LCMAX], rx[LCMAX], ry[LCMAX], rz[LCMAX];
    po_f_sub(f, h, u2, x1);
    po_f_sqr(f, hh, h);
    po_f_add(f, i, hh, hh);
    po_f_add(f, i, i, i);
    po_f_mul(f, j, h, i);
    po_f_sub(f, rr, s2, y1);
    po_f_add(f, rr, rr, rr);
    po_f_mul(f, v, x1, i);
    po_f_sqr(f, rx, rr);
    po_f_sub(f, rx, rx, j);
    po_f_sub(f, rx, rx, v);
    po_f_sub(f, rx, rx, v);
    po_f_mul(f, j, y1, j);
    po_f_add(f, j, j, j);
    po_f_sub(f, ry, v, rx);
    po_f_mul(f, ry, ry, rr);
    po_f_sub(f, ry, ry, j);
    po_f_add(f, rz, z1, h);
    po_f_sqr(f, rz, rz);
    po_f_sub(f, rz, rz, z1z1);
    po_f_sub(f, rz, rz, hh);
    memcpy(r, rx, lc * 4);
    memcpy(r + lc, ry, lc * 4);
    memcpy(r + 2 * lc, rz, lc * 4);
}

void po_to_affine(int curve, u32 *aff, const u32 *jac)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    if (po_f_is_zero(f, jac + 2 * lc)) {
        memset(aff, 0, lc * 4);
        memcpy(aff + lc, f->one, lc * 4);
        return;
    }
    u32 zi[LCMAX], zi2[LCMAX], t[LCMAX], x[LCMAX], y[LCMAX];
    po_f_inv(f, zi, jac + 2 * lc);
    po_f_sqr(f, zi2, zi);
    po_f_mul(f, x, jac, zi2);
    po_f_mul(f, t, zi, zi2);
    po_f_mul(f, y, jac + lc, t);
    memcpy(aff, x, lc * 4);
    memcpy(aff + lc, y, lc * 4);
}

void po_to_projective(int curve, u32 *hom, const u32 *jac)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    u32 x[LCMAX], y[LCMAX], z[LCMAX];
    po_f_mul(f, x, jac, jac + 2 * lc);
    memcpy(y, jac + lc, lc * 4);
    po_f_sqr(f, z, jac + 2 * lc);
    po_f_mul(f, z, z, jac + 2 * lc);
    memcpy(hom, x, lc * 4);
    memcpy(hom + lc, y, lc * 4);
    memcpy(hom + 2 * lc, z, lc * 4);
}

void po_hom_to_affine(int curve, u32 *aff, const u32 *hom)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    if (po_f_is_zero(f, hom + 2 * lc)) {
        memset(aff, 0, lc * 4);
        memcpy(aff + lc, f->one, lc * 4);
        return;
    }
    u32 zi[LCMAX], x[LCMAX], y[LCMAX];
    po_f_inv(f, zi, hom + 2 * lc);
    po_f_mul(f, x, hom, zi);
    po_f_mul(f, y, hom + lc, zi);
    memcpy(aff, x, lc * 4);
    memcpy(aff + lc, y, lc * 4);
}

void po_scalar_mul(int curve, u32 *jac, const u32 *aff, const u32 *k, unsigned nlimbs)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    u32 acc[3 * LCMAX];
    memset(acc, 0, sizeof acc);
    for (int bit = (int)nlimbs * 32 - 1; bit >= 0; bit--) {
        po_dbl(curve, acc, acc);
        if ((k[bit >> 5] >> (bit & 31)) & 1) po_madd(curve, acc, acc, aff);
    }
    memcpy(jac, acc, 3 * lc * 4);
}

int po_curve_vec(int curve, int op, u32 *r, const u32 *a, const u32 *b, size_t n)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    for (size_t i = 0; i < n; i++) {
        switch (op) {
        case PO_COP_MADD: po_madd(curve, r + i * 3 * lc, a + i * 3 * lc, b + i * 2 * lc); break;
        case PO_COP_ADD: po_add(curve, r + i * 3 * lc, a + i * 3 * lc, b + i * 3 * lc); break;
        case PO_COP_DBL: po_dbl(curve, r + i * 3 * lc, a + i * 3 * lc); break;
        default: return 1;
        }
    }
    return 0;
}
