/*
 * msm.c -- CPU ORACLE (test infrastructure, see panda_oracle.h): single-threaded Pippenger
 * restating the reference's CPU "host-debug" path, src/cuda/core/unit/msm/msm_host.cuh:
 *
 *   scalar de-Montgomery      msm_host.cuh:293-296   (done on a copy here, never in place)
 *   get_slice_bit             msm_host.cuh:237-246   window widths: c ... c, last = BC-(W-1)c
 *   get_slice/calc_all_slices msm_host.cuh:50-113    digit_w = bits [w*c, min((w+1)c, BC))
 *   aggregate_buckets         msm_host.cuh:134-191   bucket[w][digit-1] += base  (madd), digit 0 skipped
 *   calc_groups               msm_host.cuh:193-213   running sum from the top bucket down
 *   calc_groups_sums          msm_host.cuh:215-235   Horner over windows, top window first
 *
 * With window_bits = 16 (BIT_S, msm_config.cuh:7) the sequence of group operations is the
 * reference's, so the Jacobian triple written to `result` equals the reference host path's.
 */
#include "panda_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef uint32_t u32;
typedef uint64_t u64;

static unsigned get_slice(const u32 *s, unsigned lo, unsigned width)
{
    /* bits [lo, lo+width) of a little-endian multi-limb integer; width <= 24 */
    unsigned m = lo >> 5, sh = lo & 31;
    u64 v = s[m];
    if (sh + width > 32) v |= (u64)s[m + 1] << 32;
    return (unsigned)((v >> sh) & ((1u << width) - 1));
}

typedef struct {
    int curve;
    const u32 *bases;
    const u32 *canon; /* canonical scalars, n x frlc */
    u64 n;
    unsigned c, W, bc, frlc, lc;
    u32 *groups; /* W Jacobian points */
    unsigned w_begin, w_end;
    int rc;
} window_job;

/* aggregate_buckets + calc_groups for windows [w_begin, w_end) */
static void *window_worker(void *arg)
{
    window_job *j = (window_job *)arg;
    const unsigned lc = j->lc, c = j->c;
    const size_t nb = ((size_t)1 << c) - 1; /* buckets_num_in_each_group, msm_host.cuh:120 */
    u32 *buckets = (u32 *)calloc(nb, 3 * lc * 4);
    if (!buckets) {
        j->rc = 2;
        return NULL;
    }
    for (unsigned w = j->w_begin; w < j->w_end; w++) {
        if (w != j->w_begin) memset(buckets, 0, nb * 3 * lc * 4);
        unsigned lo = w * c;
        unsigned width = (w < j->W - 1) ? c : (j->bc - lo);
        for (u64 i = 0; i < j->n; i++) {
            unsigned d = get_slice(j->canon + i * j->frlc, lo, width);
            if (!d) continue;
            u32 *b = buckets + (size_t)(d - 1) * 3 * lc;
            po_madd(j->curve, b, b, j->bases + i * 2 * lc);
        }
        u32 running[3 * PO_MAX_LC], sum[3 * PO_MAX_LC];
        memset(running, 0, sizeof running);
        memset(sum, 0, sizeof sum);
        for (size_t k = 0; k < nb; k++) {
            po_add(j->curve, running, running, buckets + (nb - 1 - k) * 3 * lc);
            po_add(j->curve, sum, sum, running);
        }
        memcpy(j->groups + (size_t)w * 3 * lc, sum, 3 * lc * 4);
    }
    free(buckets);
    j->rc = 0;
    return NULL;
}

int po_msm_mt(int curve, const void *bases, const void *scalars, u64 n, unsigned c, unsigned threads, void *result)
{
    const po_field *fq = po_curve_fq(curve), *fr = po_curve_fr(curve);
    if (!fq || !fr || c < 2 || c > 22) return 1;
    const unsigned lc = fq->lc, frlc = fr->lc, bc = fr->bits;
    const unsigned W = (bc + c - 1) / c; /* get_group_number, msm_host.cuh:37-41 */
    u32 *canon = (u32 *)malloc((size_t)(n ? n : 1) * frlc * 4 + 8);
    u32 *groups = (u32 *)calloc(W, 3 * lc * 4);
    if (!canon || !groups) {
        free(canon);
        free(groups);
        return 2;
    }
    for (u64 i = 0; i < n; i++) po_f_from_mont(fr, canon + i * frlc, (const u32 *)scalars + i * frlc);
    memset(canon + n * frlc, 0, 8);

    if (threads < 1) threads = 1;
    if (threads > W) threads = W;
    window_job jobs[32];
    pthread_t tids[32];
    if (threads > 32) threads = 32;
    int rc = 0;
    for (unsigned t = 0; t < threads; t++) {
        window_job *j = &jobs[t];
        j->curve = curve;
        j->bases = (const u32 *)bases;
        j->canon = canon;
        j->n = n;
        j->c = c;
        j->W = W;
        j->bc = bc;
        j->frlc = frlc;
        j->lc = lc;
        j->groups = groups;
        j->w_begin = (unsigned)((u64)W * t / threads);
        j->w_end = (unsigned)((u64)W * (t + 1) / threads);
        j->rc = 0;
    }
    if (threads == 1) {
        window_worker(&jobs[0]);
        rc = jobs[0].rc;
    } else {
        for (unsigned t = 0; t < threads; t++) pthread_create(&tids[t], NULL, window_worker, &jobs[t]);
        for (unsigned t = 0; t < threads; t++) {
            pthread_join(tids[t], NULL);
            if (jobs[t].rc) rc = jobs[t].rc;
        }
    }
    if (!rc) {
        /* calc_groups_sums: every window below the top one is c bits wide */
        u32 acc[3 * PO_MAX_LC];
        memset(acc, 0, sizeof acc);
        for (unsigned i = 0; i + 1 < W; i++) {
            po_add(curve, acc, acc, groups + (size_t)(W - 1 - i) * 3 * lc);
            for (unsigned k = 0; k < c; k++) po_dbl(curve, acc, acc);
        }
        po_add(curve, acc, acc, groups);
        memcpy(result, acc, 3 * lc * 4);
    }
    free(canon);
    free(groups);
    return rc;
}

int po_msm(int curve, const void *bases, const void *scalars, u64 n, unsigned c, void *result)
{
    return po_msm_mt(curve, bases, scalars, n, c, 1, result);
}

int po_msm_naive(int curve, const void *bases, const void *scalars, u64 n, void *result)
{
    const po_field *fq = po_curve_fq(curve), *fr = po_curve_fr(curve);
    const unsigned lc = fq->lc, frlc = fr->lc;
    u32 acc[3 * PO_MAX_LC], t[3 * PO_MAX_LC], k[PO_MAX_LC];
    memset(acc, 0, sizeof acc);
    for (u64 i = 0; i < n; i++) {
        po_f_from_mont(fr, k, (const u32 *)scalars + i * frlc);
        const u32 *b = (const u32 *)bases + i * 2 * lc;
        if (po_f_is_zero(fq, b)) continue; /* affine identity, affine.cuh:72-75 */
        po_scalar_mul(curve, t, b, k, frlc);
        po_add(curve, acc, acc, t);
    }
    memcpy(result, acc, 3 * lc * 4);
    return 0;
}
