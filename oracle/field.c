/*
 * field.c -- CPU ORACLE (test infrastructure, see panda_oracle.h): Montgomery prime-field
 * arithmetic on 32-bit limbs, restating the reference's *host* field path
 * (src/cuda/core/field/field_host.cuh) in plain C.
 *
 *   add / sub          field_host.cuh:63-160   (add then conditional subtract; borrow -> add p)
 *   mul                field_host.cuh:162-210 (schoolbook mul_limbs) + :308-382 (mont_limbs)
 *   to/from Montgomery field.cuh:566-619       (mul by R^2 / by 1)
 *   inverse            field_host.cuh:404-472  (binary extended GCD, b seeded with R^2)
 *
 * Every result is fully reduced to [0, p) (panda_reduce, field.cuh:115-121), so values are
 * unique and any correct implementation is byte-identical with the reference's.
 * Moduli: bn254/paramter.cuh:19-26 (Fq), :135-142 (Fr); bls12_377/paramter.cuh:23-37 (Fq),
 * :138-148 (Fr).  R, R^2 and -p^-1 are derived here from p and cross-checked in the tests
 * against the reference's tables (bn254/paramter.cuh:99-119,216-237).
 */
#include "panda_oracle.h"

#include <string.h>

typedef uint32_t u32;
typedef uint64_t u64;

static po_field g_fields[PO_NUM_FIELDS] = {
    /* BN254 Fq */
    {8, 254, 0, {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}, {0}, {0}},
    /* BN254 Fr */
    {8, 254, 0, {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}, {0}, {0}},
    /* BLS12-377 Fq */
    {12, 377, 0, {0x00000001u, 0x8508c000u, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu, 0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu, 0xc63b05c0u, 0x17c510eau, 0x01ae3a46u}, {0}, {0}},
    /* BLS12-377 Fr */
    {8, 253, 0, {0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu, 0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu}, {0}, {0}},
    /* BLS12-381 Fq, Fr: the reference only names the curve (curve.cuh:12); standard parameters, parity unpinned by the reference */
    {12, 381, 0, {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau}, {0}, {0}},
    {8, 255, 0, {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u}, {0}, {0}},
};
static int g_fields_ready = 0;

static int ge(const u32 *a, const u32 *b, unsigned lc)
{
    for (int i = (int)lc - 1; i >= 0; i--) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return 1;
}

static u32 add_n(u32 *r, const u32 *a, const u32 *b, unsigned lc)
{
    u64 c = 0;
    for (unsigned i = 0; i < lc; i++) {
        c += (u64)a[i] + b[i];
        r[i] = (u32)c;
        c >>= 32;
    }
    return (u32)c;
}

static u32 sub_n(u32 *r, const u32 *a, const u32 *b, unsigned lc)
{
    u64 br = 0;
    for (unsigned i = 0; i < lc; i++) {
        u64 d = (u64)a[i] - b[i] - br;
        r[i] = (u32)d;
        br = (d >> 32) & 1;
    }
    return (u32)br;
}

/* r = 2a mod p for a < p (used only to derive R and R^2) */
static void dbl_mod(const po_field *f, u32 *r, const u32 *a)
{
    u32 t[PO_MAX_LC];
    u32 c = add_n(t, a, a, f->lc);
    if (c || ge(t, f->p, f->lc)) sub_n(t, t, f->p, f->lc);
    memcpy(r, t, f->lc * 4);
}

static void fields_init(void)
{
    if (g_fields_ready) return;
    for (int k = 0; k < PO_NUM_FIELDS; k++) {
        po_field *f = &g_fields[k];
        /* -p^-1 mod 2^32 by Newton iteration */
        u32 p0 = f->p[0], x = 1;
        for (int i = 0; i < 6; i++) x *= 2u - p0 * x;
        f->inv = (u32)(0u - x);
        /* R = 2^(32 lc) mod p, R^2 = 2^(64 lc) mod p by repeated doubling of 1 */
        u32 t[PO_MAX_LC] = {1};
        for (unsigned i = 0; i < 32 * f->lc; i++) dbl_mod(f, t, t);
        memcpy(f->one, t, sizeof t);
        for (unsigned i = 0; i < 32 * f->lc; i++) dbl_mod(f, t, t);
        memcpy(f->r2, t, sizeof t);
    }
    g_fields_ready = 1;
}

const po_field *po_field_get(int id)
{
    fields_init();
    if (id < 0 || id >= PO_NUM_FIELDS) return NULL;
    return &g_fields[id];
}
const po_field *po_curve_fq(int curve) { return po_field_get(2 * curve); }     /* field ids are (Fq, Fr) pairs in curve order */
const po_field *po_curve_fr(int curve) { return po_field_get(2 * curve + 1); }

void po_f_add(const po_field *f, u32 *r, const u32 *a, const u32 *b)
{
    u32 t[PO_MAX_LC];
    u32 c = add_n(t, a, b, f->lc);
    if (c || ge(t, f->p, f->lc)) sub_n(t, t, f->p, f->lc);
    memcpy(r, t, f->lc * 4);
}

void po_f_sub(const po_field *f, u32 *r, const u32 *a, const u32 *b)
{
    u32 t[PO_MAX_LC];
    if (sub_n(t, a, b, f->lc)) add_n(t, t, f->p, f->lc);
    memcpy(r, t, f->lc * 4);
}

void po_f_neg(const po_field *f, u32 *r, const u32 *a)
{
    u32 z[PO_MAX_LC] = {0};
    po_f_sub(f, r, z, a);
}

/* wide = a*b (schoolbook), then lc rounds of Montgomery reduction, then one conditional subtract */
void po_f_mul(const po_field *f, u32 *r, const u32 *a, const u32 *b)
{
    const unsigned lc = f->lc;
    u32 w[2 * PO_MAX_LC + 1];
    memset(w, 0, sizeof w);
    for (unsigned i = 0; i < lc; i++) {
        u64 c = 0;
        for (unsigned j = 0; j < lc; j++) {
            c += (u64)a[i] * b[j] + w[i + j];
            w[i + j] = (u32)c;
            c >>= 32;
        }
        w[i + lc] = (u32)c;
    }
    u32 carry2 = 0;
    for (unsigned i = 0; i < lc; i++) {
        u32 k = w[i] * f->inv;
        u64 c = 0;
        for (unsigned j = 0; j < lc; j++) {
            c += (u64)k * f->p[j] + w[i + j];
            w[i + j] = (u32)c;
            c >>= 32;
        }
        c += (u64)w[i + lc] + carry2;
        w[i + lc] = (u32)c;
        carry2 = (u32)(c >> 32);
    }
    u32 *hi = w + lc;
    if (carry2 || ge(hi, f->p, lc)) sub_n(hi, hi, f->p, lc);
    memcpy(r, hi, lc * 4);
}

void po_f_sqr(const po_field *f, u32 *r, const u32 *a) { po_f_mul(f, r, a, a); }

void po_f_to_mont(const po_field *f, u32 *r, const u32 *a) { po_f_mul(f, r, a, f->r2); }

void po_f_from_mont(const po_field *f, u32 *r, const u32 *a)
{
    u32 one[PO_MAX_LC] = {1};
    po_f_mul(f, r, a, one);
}

int po_f_is_zero(const po_field *f, const u32 *a)
{
    u32 acc = 0;
    for (unsigned i = 0; i < f->lc; i++) acc |= a[i];
    return acc == 0;
}

int po_f_eq(const po_field *f, const u32 *a, const u32 *b) { return memcmp(a, b, f->lc * 4) == 0; }

static void rshift1(u32 *a, unsigned lc, u32 top)
{
    for (unsigned i = 0; i + 1 < lc; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 31);
    a[lc - 1] = (a[lc - 1] >> 1) | (top << 31);
}

static int is_one_limbs(const u32 *a, unsigned lc)
{
    if (a[0] != 1) return 0;
    for (unsigned i = 1; i < lc; i++)
        if (a[i]) return 0;
    return 1;
}

/* binary extended GCD; x = aR in, (aR)^-1 * R^2 = a^-1 R out.  Zero maps to zero. */
void po_f_inv(const po_field *f, u32 *r, const u32 *a)
{
    const unsigned lc = f->lc;
    if (po_f_is_zero(f, a)) {
        memset(r, 0, lc * 4);
        return;
    }
    u32 u[PO_MAX_LC], v[PO_MAX_LC], b[PO_MAX_LC], c[PO_MAX_LC];
    memcpy(u, a, lc * 4);
    memcpy(v, f->p, lc * 4);
    memcpy(b, f->r2, lc * 4);
    memset(c, 0, lc * 4);
    while (!is_one_limbs(u, lc) && !is_one_limbs(v, lc)) {
        while ((u[0] & 1) == 0) {
            rshift1(u, lc, 0);
            u32 top = 0;
            if (b[0] & 1) top = add_n(b, b, f->p, lc);
            rshift1(b, lc, top);
        }
        while ((v[0] & 1) == 0) {
            rshift1(v, lc, 0);
            u32 top = 0;
            if (c[0] & 1) top = add_n(c, c, f->p, lc);
            rshift1(c, lc, top);
        }
        if (ge(u, v, lc)) {
            sub_n(u, u, v, lc);
            po_f_sub(f, b, b, c);
        } else {
            sub_n(v, v, u, lc);
            po_f_sub(f, c, c, b);
        }
    }
    memcpy(r, is_one_limbs(u, lc) ? b : c, lc * 4);
}

void po_f_pow_u64(const po_field *f, u32 *r, const u32 *a, u64 e)
{
    u32 acc[PO_MAX_LC], base[PO_MAX_LC];
    memcpy(acc, f->one, sizeof acc);
    memcpy(base, a, f->lc * 4);
    while (e) {
        if (e & 1) po_f_mul(f, acc, acc, base);
        po_f_sqr(f, base, base);
        e >>= 1;
    }
    memcpy(r, acc, f->lc * 4);
}

int po_f_vec(int field_id, int op, u32 *r, const u32 *a, const u32 *b, size_t n)
{
    const po_field *f = po_field_get(field_id);
    if (!f) return 1;
    const unsigned lc = f->lc;
    for (size_t i = 0; i < n; i++) {
        u32 *ri = r + i * lc;
        const u32 *ai = a + i * lc;
        const u32 *bi = b ? b + i * lc : NULL;
        switch (op) {
        case PO_OP_ADD: po_f_add(f, ri, ai, bi); break;
        case PO_OP_SUB: po_f_sub(f, ri, ai, bi); break;
        case PO_OP_MUL: po_f_mul(f, ri, ai, bi); break;
        case PO_OP_SQR: po_f_sqr(f, ri, ai); break;
        case PO_OP_TO_MONT: po_f_to_mont(f, ri, ai); break;
        case PO_OP_FROM_MONT: po_f_from_mont(f, ri, ai); break;
        case PO_OP_INV: po_f_inv(f, ri, ai); break;
        default: return 1;
        }
    }
    return 0;
}

int po_f_scale(int field_id, void *out, const void *in, const void *s, size_t n)
{
    const po_field *f = po_field_get(field_id);
    if (!f) return 1;
    for (size_t i = 0; i < n; i++) po_f_mul(f, (u32 *)out + i * f->lc, (const u32 *)in + i * f->lc, (const u32 *)s);
    return 0;
}
