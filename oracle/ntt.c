/*
 * ntt.c -- CPU ORACLE (test infrastructure, see panda_oracle.h): number-theoretic transform.
 *
 * PARITY UNPINNED BY THE REFERENCE: src/cuda/core/unit/ntt/fft.cu compiles its kernel bodies
 * out (`#if 0`, fft.cu:18-35, 89-101, 117-168), there is no CPU NTT in the reference and the
 * golden in/out files are missing (.MISSING_LARGE_BLOBS).  What is restated here:
 *
 *   po_dft_naive   the definition the commented-out code computes:
 *                  y[k] = sum_j x[j] * omega^(j k), natural order in and out, Montgomery in/out,
 *                  no 1/n scaling (fft.cu:103-169 read together with the pass loop :171-216)
 *   po_ntt         O(n log n) radix-2 Cooley-Tukey of the same definition
 *   po_ntt_passes  the reference's pass structure: radix-2^deg Stockham passes, deg = min(8,
 *                  remaining), src/dst ping-pong, flag = passes & 1 (fft.cu:177,193-211); the
 *                  per-pass index arithmetic follows the commented radix_fft (fft.cu:122-167)
 *   po_ntt_eval_at one output of the same definition by Horner's rule in O(n): the check for sizes
 *                  po_ntt would need minutes for (spot-check by direct evaluation)
 *   po_root_of_unity  BN254 Fr: generator 7, two-adicity 28 (bn254/paramter.cuh:241-258)
 */
#include "panda_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef uint32_t u32;
typedef uint64_t u64;
#define LCMAX PO_MAX_LC

static unsigned bitrev(unsigned v, unsigned bits)
{
    unsigned r = 0;
    for (unsigned i = 0; i < bits; i++) {
        r = (r << 1) | (v & 1);
        v >>= 1;
    }
    return r;
}

int po_dft_naive(int field_id, void *out_, const void *in_, const void *omega, unsigned log_n)
{
    const po_field *f = po_field_get(field_id);
    if (!f || log_n > 14) return 1;
    const unsigned lc = f->lc;
    const size_t n = (size_t)1 << log_n;
    const u32 *in = (const u32 *)in_;
    u32 *out = (u32 *)out_;
    u32 *pw = (u32 *)malloc(n * lc * 4); /* omega^t, t < n */
    memcpy(pw, f->one, lc * 4);
    for (size_t t = 1; t < n; t++) po_f_mul(f, pw + t * lc, pw + (t - 1) * lc, (const u32 *)omega);
    u32 *tmp = (u32 *)malloc(n * lc * 4);
    for (size_t k = 0; k < n; k++) {
        u32 acc[LCMAX] = {0}, t[LCMAX];
        for (size_t j = 0; j < n; j++) {
            po_f_mul(f, t, in + j * lc, pw + ((j * k) & (n - 1)) * lc);
            po_f_add(f, acc, acc, t);
        }
        memcpy(tmp + k * lc, acc, lc * 4);
    }
    memcpy(out, tmp, n * lc * 4);
    free(tmp);
    free(pw);
    return 0;
}

int po_ntt(int field_id, void *out_, const void *in_, const void *omega, unsigned log_n)
{
    const po_field *f = po_field_get(field_id);
    if (!f || log_n > 30) return 1;
    const unsigned lc = f->lc;
    const size_t n = (size_t)1 << log_n;
    u32 *a = (u32 *)malloc(n * lc * 4);
    const u32 *in = (const u32 *)in_;
    for (size_t i = 0; i < n; i++) memcpy(a + (size_t)bitrev((unsigned)i, log_n) * lc, in + i * lc, lc * 4);
    /* w_s = omega^(n / 2^s) for stage s */
    u32 wst[32][LCMAX];
    memcpy(wst[log_n], omega, lc * 4);
    for (int s = (int)log_n - 1; s >= 1; s--) po_f_sqr(f, wst[s], wst[s + 1]);
    for (unsigned s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, h = m >> 1;
        for (size_t k = 0; k < n; k += m) {
            u32 w[LCMAX], t[LCMAX], u[LCMAX];
            memcpy(w, f->one, lc * 4);
            for (size_t j = 0; j < h; j++) {
                u32 *x0 = a + (k + j) * lc, *x1 = a + (k + j + h) * lc;
                po_f_mul(f, t, w, x1);
                memcpy(u, x0, lc * 4);
                po_f_add(f, x0, u, t);
                po_f_sub(f, x1, u, t);
                po_f_mul(f, w, w, wst[s]);
            }
        }
    }
    memcpy(out_, a, n * lc * 4);
    free(a);
    return 0;
}

/* omega^e by square-and-multiply over the precomputed omegas[t] = omega^(2^t) (fft.cu:38-49, 86-101) */
static void pow_lookup(const po_field *f, u32 *r, u32 (*omegas)[LCMAX], unsigned e)
{
    u32 acc[LCMAX];
    memcpy(acc, f->one, sizeof acc);
    for (unsigned i = 0; e; i++, e >>= 1)
        if (e & 1) po_f_mul(f, acc, acc, omegas[i]);
    memcpy(r, acc, f->lc * 4);
}

int po_ntt_passes(int field_id, void *out_, const void *in_, const void *omega, unsigned log_n, unsigned *flag)
{
    const po_field *f = po_field_get(field_id);
    if (!f || log_n > 26) return 1;
    const unsigned lc = f->lc;
    const unsigned n = 1u << log_n;
    u32(*omegas)[LCMAX] = (u32(*)[LCMAX])malloc(32 * sizeof(u32[LCMAX]));
    memcpy(omegas[0], omega, lc * 4);
    for (unsigned t = 1; t < 32; t++) po_f_sqr(f, omegas[t], omegas[t - 1]);
    const unsigned max_deg = log_n < 8 ? log_n : 8; /* MAX_LOG2_RADIX, fft.cu:9,177 */
    /* pq[t] = (omega^(n >> max_deg))^t, t < 2^(max_deg-1)   (fft.cu:51-60) */
    unsigned npq = max_deg ? (1u << max_deg >> 1) : 1;
    if (npq == 0) npq = 1;
    u32 *pq = (u32 *)malloc((size_t)npq * lc * 4);
    {
        u32 tw[LCMAX];
        pow_lookup(f, tw, omegas, n >> max_deg);
        memcpy(pq, f->one, lc * 4);
        for (unsigned t = 1; t < npq; t++) po_f_mul(f, pq + t * lc, pq + (t - 1) * lc, tw);
    }
    u32 *src = (u32 *)malloc((size_t)n * lc * 4), *dst = (u32 *)malloc((size_t)n * lc * 4);
    memcpy(src, in_, (size_t)n * lc * 4);
    u32 *u = (u32 *)malloc(256 * lc * 4);
    unsigned log_p = 0, passes = 0;
    while (log_p < log_n) {
        unsigned deg = log_n - log_p < max_deg ? log_n - log_p : max_deg;
        unsigned p = 1u << log_p, count = 1u << deg, counth = count >> 1;
        for (unsigned blk = 0; blk < (n >> deg); blk++) {
            unsigned k = blk & (p - 1);
            const u32 *x = src + (size_t)blk * lc;
            u32 *y = dst + ((size_t)((blk - k) << deg) + k) * lc;
            u32 tw[LCMAX], t[LCMAX];
            pow_lookup(f, tw, omegas, (n >> log_p >> deg) * k);
            memcpy(t, f->one, lc * 4);
            for (unsigned i = 0; i < count; i++) {
                po_f_mul(f, u + i * lc, t, x + (size_t)i * (n >> deg) * lc);
                po_f_mul(f, t, t, tw);
            }
            const unsigned pqshift = max_deg - deg;
            for (unsigned rnd = 0; rnd < deg; rnd++) {
                unsigned bit = counth >> rnd;
                for (unsigned i = 0; i < counth; i++) {
                    unsigned di = i & (bit - 1);
                    unsigned i0 = (i << 1) - di, i1 = i0 + bit;
                    u32 a[LCMAX];
                    memcpy(a, u + i0 * lc, lc * 4);
                    po_f_add(f, u + i0 * lc, a, u + i1 * lc);
                    po_f_sub(f, u + i1 * lc, a, u + i1 * lc);
                    if (di) po_f_mul(f, u + i1 * lc, pq + ((size_t)(di << rnd << pqshift)) * lc, u + i1 * lc);
                }
            }
            for (unsigned i = 0; i < count; i++) memcpy(y + (size_t)i * p * lc, u + (size_t)bitrev(i, deg) * lc, lc * 4);
        }
        u32 *sw = src;
        src = dst;
        dst = sw;
        log_p += deg;
        passes++;
    }
    if (flag) *flag = passes & 1;
    memcpy(out_, src, (size_t)n * lc * 4);
    free(u);
    free(src);
    free(dst);
    free(pq);
    free(omegas);
    return 0;
}

int po_root_of_unity(int field_id, unsigned log_n, void *omega_mont)
{
    const po_field *f = po_field_get(field_id);
    unsigned two_adicity, gen;
    if (field_id == PO_FIELD_BN254_FR) {
        two_adicity = 28;
        gen = 7;
    } else if (field_id == PO_FIELD_BLS12_377_FR) {
        two_adicity = 47;
        gen = 22;
    } else if (field_id == PO_FIELD_BLS12_381_FR) {
        two_adicity = 32;
        gen = 7;
    } else
        return 1;
    if (log_n > two_adicity) return 1;
    const unsigned lc = f->lc;
    /* e = (p - 1) >> two_adicity */
    u32 e[LCMAX];
    memcpy(e, f->p, lc * 4);
    e[0] -= 1;
    for (unsigned s = 0; s < two_adicity; s++) {
        for (unsigned i = 0; i + 1 < lc; i++) e[i] = (e[i] >> 1) | (e[i + 1] << 31);
        e[lc - 1] >>= 1;
    }
    u32 g[LCMAX] = {0}, gm[LCMAX], acc[LCMAX];
    g[0] = gen;
    po_f_to_mont(f, gm, g);
    memcpy(acc, f->one, lc * 4);
    for (int bit = (int)lc * 32 - 1; bit >= 0; bit--) {
        po_f_sqr(f, acc, acc);
        if ((e[bit >> 5] >> (bit & 31)) & 1) po_f_mul(f, acc, acc, gm);
    }
    for (unsigned s = log_n; s < two_adicity; s++) po_f_sqr(f, acc, acc);
    memcpy(omega_mont, acc, lc * 4);
    return 0;
}

/* y[k] = sum_j x[j] * (omega^k)^j for ONE output index k, by Horner's rule from the top coefficient down:
 * n multiplications and n additions, no table.  Same definition as po_dft_naive (fft.cu:103-169). */
int po_ntt_eval_at(int field_id, void *out_, const void *in_, const void *omega, unsigned log_n, uint64_t k)
{
    const po_field *f = po_field_get(field_id);
    if (!f || log_n > 30) return 1;
    const unsigned lc = f->lc;
    const size_t n = (size_t)1 << log_n;
    const u32 *in = (const u32 *)in_;
    u32 wk[LCMAX], acc[LCMAX] = {0}, t[LCMAX];
    po_f_pow_u64(f, wk, (const u32 *)omega, k & (n - 1));
    for (size_t j = n; j-- > 0;) {
        po_f_mul(f, t, acc, wk);
        po_f_add(f, acc, t, in + j * lc);
    }
    memcpy(out_, acc, lc * 4);
    return 0;
}
