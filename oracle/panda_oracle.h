/*
 * panda_oracle.h -- CPU ORACLE for the MSM + NTT hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is a plain-C restatement of the reference's CPU "host-debug" algorithm
 * (JasonHopeSpace/panda, src/cuda/core/unit/msm/msm_host.cuh and the field/curve headers
 * below it).  It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg can check / time the HIP path against an independent CPU answer.
 * Nothing under panda_amd/ (the product) links, imports, calls or executes it.
 *
 * Pinning: the restatement is checked against the reference's own golden vector
 * (src/cuda/test/data/msm/k13/{bases,scalars,result_affine}.bin, committed as data under
 * tests/golden/) and against an independent pure-Python big-integer computation
 * (tests/pyref.py).  The reference host path itself is NOT built here: it needs
 * <cuda_runtime.h>, which this image lacks, and stand-in headers are not allowed.
 * NTT: the reference has no runnable NTT (kernel bodies are `#if 0`, fft.cu:18-35,89-101,
 * 117-168) and no NTT fixture, so NTT parity is "unpinned by the reference"; the NTT here
 * follows the definition the commented-out code implements (y[k] = sum_j x[j] w^(jk)) and is
 * pinned by the O(n^2) DFT, round trips and linearity only.
 * BLS12-381 (curve id 2): the reference names the curve (curve.cuh:12) but holds no parameters,
 * code or fixtures for it, so it too is "parity unpinned by the reference"; the standard
 * parameters used here are pinned by tests/pyref.py (generator on-curve, r*G = O, group law, MSM).
 *
 * Wire formats (reference src/utils.rs:1-14, field_storage.cuh:12-16, affine.cuh:11-19,
 * projective.cuh:9-20):
 *   field element  : LC little-endian u32 limbs, Montgomery form (x * 2^(32*LC) mod p)
 *   scalar         : Fr element, 32 bytes
 *   affine base    : x || y            (BN254 64 B, BLS12-377 96 B); identity <=> x == 0
 *   result         : X || Y || Z       Jacobian (x = X/Z^2, y = Y/Z^3); identity <=> Z == 0
 */
#ifndef PANDA_ORACLE_H
#define PANDA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_MAX_LC 12

enum { PO_CURVE_BN254 = 0, PO_CURVE_BLS12_377 = 1, PO_CURVE_BLS12_381 = 2 };
enum { PO_FIELD_BN254_FQ = 0, PO_FIELD_BN254_FR = 1, PO_FIELD_BLS12_377_FQ = 2, PO_FIELD_BLS12_377_FR = 3, PO_FIELD_BLS12_381_FQ = 4, PO_FIELD_BLS12_381_FR = 5 };
#define PO_NUM_FIELDS 6
#define PO_NUM_CURVES 3

typedef struct po_field {
    uint32_t lc;               /* limbs */
    uint32_t bits;             /* bit length of the modulus */
    uint32_t inv;              /* -p^-1 mod 2^32 */
    uint32_t p[PO_MAX_LC];     /* modulus */
    uint32_t one[PO_MAX_LC];   /* R mod p */
    uint32_t r2[PO_MAX_LC];    /* R^2 mod p */
} po_field;

/* field table; constants (one, r2, inv) are derived from p at first use */
const po_field *po_field_get(int field_id);
const po_field *po_curve_fq(int curve);
const po_field *po_curve_fr(int curve);

/* field arithmetic, all values fully reduced, Montgomery form where it matters */
void po_f_add(const po_field *f, uint32_t *r, const uint32_t *a, const uint32_t *b);
void po_f_sub(const po_field *f, uint32_t *r, const uint32_t *a, const uint32_t *b);
void po_f_neg(const po_field *f, uint32_t *r, const uint32_t *a);
void po_f_mul(const po_field *f, uint32_t *r, const uint32_t *a, const uint32_t *b);
void po_f_sqr(const po_field *f, uint32_t *r, const uint32_t *a);
void po_f_inv(const po_field *f, uint32_t *r, const uint32_t *a);
void po_f_to_mont(const po_field *f, uint32_t *r, const uint32_t *a);
void po_f_from_mont(const po_field *f, uint32_t *r, const uint32_t *a);
int po_f_is_zero(const po_field *f, const uint32_t *a);
int po_f_eq(const po_field *f, const uint32_t *a, const uint32_t *b);
void po_f_pow_u64(const po_field *f, uint32_t *r, const uint32_t *a, uint64_t e);

/* vectorised helpers used by the element-wise parity tests: op over n elements */
enum { PO_OP_ADD = 0, PO_OP_SUB = 1, PO_OP_MUL = 2, PO_OP_SQR = 3, PO_OP_TO_MONT = 4, PO_OP_FROM_MONT = 5, PO_OP_INV = 6 };
int po_f_vec(int field_id, int op, uint32_t *r, const uint32_t *a, const uint32_t *b, size_t n);

/* curve ops on raw limb buffers: Jacobian = 3*LC limbs, affine = 2*LC limbs */
void po_madd(int curve, uint32_t *r, const uint32_t *p1, const uint32_t *aff);
void po_add(int curve, uint32_t *r, const uint32_t *p1, const uint32_t *p2);
void po_dbl(int curve, uint32_t *r, const uint32_t *p1);
void po_to_affine(int curve, uint32_t *aff, const uint32_t *jac);
void po_to_projective(int curve, uint32_t *hom, const uint32_t *jac);
/* homogeneous (x = X/Z, y = Y/Z) -> affine, for checking PROJECTIVE-coordinate results */
void po_hom_to_affine(int curve, uint32_t *aff, const uint32_t *hom);
int po_is_on_curve(int curve, const uint32_t *aff);
void po_generator(int curve, uint32_t *aff);
/* k*P for a canonical (non-Montgomery) little-endian scalar of nlimbs u32 */
void po_scalar_mul(int curve, uint32_t *jac, const uint32_t *aff, const uint32_t *k, unsigned nlimbs);
int po_curve_vec(int curve, int op, uint32_t *r, const uint32_t *a, const uint32_t *b, size_t n);
enum { PO_COP_MADD = 0, PO_COP_ADD = 1, PO_COP_DBL = 2 };

/*
 * Pippenger MSM, restating msm_host.cuh:267-370.  `scalars` are Montgomery-form Fr and are
 * NOT modified (the reference converts in place, msm_host.cuh:293-296; see SURVEY bug #3).
 * window_bits = 16 reproduces the reference (BIT_S, msm_config.cuh:7); other values give the
 * same group element (tests use smaller windows to stay fast).  Result: Jacobian X||Y||Z.
 */
int po_msm(int curve, const void *bases, const void *scalars, uint64_t n, unsigned window_bits, void *result_jacobian);
/* naive sum of double-and-add products, independent of the bucket method */
int po_msm_naive(int curve, const void *bases, const void *scalars, uint64_t n, void *result_jacobian);
/* same algorithm, bucket work split over `threads` pthreads by window (for cpu_baseline only) */
int po_msm_mt(int curve, const void *bases, const void *scalars, uint64_t n, unsigned window_bits, unsigned threads, void *result_jacobian);

/*
 * NTT over BN254 Fr (field id selectable): natural order in, natural order out,
 * y[k] = sum_j x[j] * omega^(j*k), Montgomery form in/out, no scaling
 * (definition implemented by the commented-out radix_fft, fft.cu:103-169, and its pass loop :171-216).
 */
int po_ntt(int field_id, void *out, const void *in, const void *omega, unsigned log_n);
int po_dft_naive(int field_id, void *out, const void *in, const void *omega, unsigned log_n);
/* one output y[k] of the same definition in O(n) (Horner) */
int po_ntt_eval_at(int field_id, void *out, const void *in, const void *omega, unsigned log_n, uint64_t k);
/* literal restatement of the reference's pass structure (radix-2^deg Stockham passes with ping-pong);
 * returns the `flag` the reference would write (fft.cu:211), result always copied to `out` */
int po_ntt_passes(int field_id, void *out, const void *in, const void *omega, unsigned log_n, unsigned *flag);
/* omega of order 2^log_n for the field (BN254 Fr: 7^((r-1)/2^28) as bn254/paramter.cuh:241-258) */
int po_root_of_unity(int field_id, unsigned log_n, void *omega_mont);
/* out[i] = a[i] * s (Montgomery), used for the n^-1 scaling of inverse transforms */
int po_f_scale(int field_id, void *out, const void *in, const void *s, size_t n);

/*
 * Synthetic inputs.  Deterministic in (seed, index), identical to the HIP
 * generator kernels (panda_amd/csrc/gen.hip) so that host and device can build the same data.
 *   scalars : raw uniform values in [0, r) by rejection of 32*LC-bit splitmix64 words masked to
 *             the modulus bit length; interpreted as Montgomery-form wire scalars.
 *   bases   : P_i = m_i * G, m_i = splitmix64(seed ^ GOLD*(i+1)) | 1, affine Montgomery.
 */
void po_gen_scalars(int field_id, uint64_t seed, uint64_t first, uint64_t n, void *out);
uint64_t po_gen_multiplier(uint64_t seed, uint64_t i);
int po_gen_bases(int curve, uint64_t seed, uint64_t first, uint64_t n, void *out);
/* acc = sum_i from_mont(s_i) * m_i mod r  (canonical, LC limbs) for the linearity check */
int po_linear_combination(int curve, uint64_t seed_bases, uint64_t first, const void *scalars, uint64_t n, void *acc_canonical);

#ifdef __cplusplus
}
#endif
#endif
