/*
 * gen.c -- CPU ORACLE (test infrastructure, see panda_oracle.h): deterministic synthetic
 * inputs and the O(n) linearity check used at sizes no CPU MSM reaches.
 * The HIP generator kernels (panda_amd/csrc/gen.hip) implement the same functions of
 * (seed, index); tests assert the two agree byte for byte.
 *
 * The reference has no input generator of its own beyond `G::Projective::rand` /
 * `ScalarField::rand` in tests/test.rs:19-46 (unseeded ChaCha20); this replaces it with a
 * seedable counter-based one.
 */
#include "panda_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef uint32_t u32;
typedef uint64_t u64;
#define LCMAX PO_MAX_LC

static inline u64 splitmix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static int lt(const u32 *a, const u32 *b, unsigned lc)
{
    for (int i = (int)lc - 1; i >= 0; i--)
        if (a[i] != b[i]) return a[i] < b[i];
    return 0;
}

void po_gen_scalars(int field_id, u64 seed, u64 first, u64 n, void *out_)
{
    const po_field *f = po_field_get(field_id);
    const unsigned lc = f->lc;
    u32 *out = (u32 *)out_;
    const unsigned top_bits = f->bits - 32 * (lc - 1);
    const u32 top_mask = top_bits >= 32 ? 0xffffffffu : ((1u << top_bits) - 1);
    for (u64 i = 0; i < n; i++) {
        u64 base = seed ^ ((first + i) * 0xD1342543DE82EF95ull);
        u32 v[LCMAX];
        unsigned attempt = 0;
        for (;;) {
            for (unsigned k = 0; k < lc / 2; k++) {
                u64 w = splitmix64(base + (u64)attempt * (lc / 2) + k);
                v[2 * k] = (u32)w;
                v[2 * k + 1] = (u32)(w >> 32);
            }
            v[lc - 1] &= top_mask;
            if (lt(v, f->p, lc)) break;
            if (++attempt == 64) { /* probability 2^-128: clear the top bit, always < p */
                v[lc - 1] &= top_mask >> 1;
                break;
            }
        }
        memcpy(out + i * lc, v, lc * 4);
    }
}

u64 po_gen_multiplier(u64 seed, u64 i) { return splitmix64(seed ^ (0x9E3779B97F4A7C15ull * (i + 1))) | 1ull; }

/* fixed-base table T[j][d-1] = d * 256^j * G (affine), j < 8, d in 1..255 */
static u32 *g_table[PO_NUM_CURVES] = {NULL};

static const u32 *fixed_base_table(int curve)
{
    if (g_table[curve]) return g_table[curve];
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    u32 *t = (u32 *)malloc((size_t)8 * 255 * 2 * lc * 4);
    u32 base[2 * LCMAX], jac[3 * LCMAX];
    po_generator(curve, base);
    for (unsigned j = 0; j < 8; j++) {
        memset(jac, 0, sizeof jac);
        for (unsigned d = 1; d <= 255; d++) {
            po_madd(curve, jac, jac, base);
            po_to_affine(curve, t + ((size_t)j * 255 + (d - 1)) * 2 * lc, jac);
        }
        /* base <- 256 * base */
        po_madd(curve, jac, jac, base);
        po_to_affine(curve, base, jac);
    }
    g_table[curve] = t;
    return t;
}

int po_gen_bases(int curve, u64 seed, u64 first, u64 n, void *out_)
{
    const po_field *f = po_curve_fq(curve);
    const unsigned lc = f->lc;
    const u32 *tab = fixed_base_table(curve);
    u32 *out = (u32 *)out_;
    for (u64 i = 0; i < n; i++) {
        u64 m = po_gen_multiplier(seed, first + i);
        u32 jac[3 * LCMAX];
        memset(jac, 0, sizeof jac);
        for (unsigned j = 0; j < 8; j++) {
            unsigned d = (unsigned)(m >> (8 * j)) & 255;
            if (d) po_madd(curve, jac, jac, tab + ((size_t)j * 255 + (d - 1)) * 2 * lc);
        }
        po_to_affine(curve, out + i * 2 * lc, jac);
    }
    return 0;
}

int po_linear_combination(int curve, u64 seed_bases, u64 first, const void *scalars, u64 n, void *acc_)
{
    const po_field *fr = po_curve_fr(curve);
    const unsigned lc = fr->lc;
    u32 acc[LCMAX] = {0};
    for (u64 i = 0; i < n; i++) {
        u64 m = po_gen_multiplier(seed_bases, first + i);
        u32 ml[LCMAX] = {0}, t[LCMAX];
        ml[0] = (u32)m;
        ml[1] = (u32)(m >> 32);
        /* mont_mul(s*R, m) = s*m: canonical product of the de-Montgomeryed scalar and m */
        po_f_mul(fr, t, (const u32 *)scalars + i * lc, ml);
        po_f_add(fr, acc, acc, t);
    }
    memcpy(acc_, acc, lc * 4);
    return 0;
}
