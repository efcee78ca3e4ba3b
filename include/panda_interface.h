/*
 * panda_interface.h -- the drop-in C ABI of the MI355X-native MSM + NTT library.
 *
 * Every declaration in part 1 replaces, symbol for symbol and struct layout for struct layout,
 * an `extern "C"` entry point of the reference's L2 shim so that the reference's Rust `gpu_ffi`
 * (src/gpu_ffi/binding.rs:3-115) binds to this library unchanged:
 *
 *   reference declaration            src/cuda/core/panda_interface.cuh:10-110
 *   reference definition             src/cuda/core/panda_interface.cu:11-191
 *   Rust-side extern block           src/gpu_ffi/binding.rs:3-115
 *   Rust-side repr(C) structs        src/gpu_ffi/common.rs:40-44,89-93,134-138,160-208
 *
 * Part 2 are the four symbols the Rust side declares but the reference never defines
 * (binding.rs:14,16,54-56).  Part 3 is additive (no reference counterpart): BLS12-377 / BLS12-381 / BN254 G2,
 * cached-base registration and tables, the in-call upload pipeline, inverse / coset / bit-reversed NTTs, multi-GPU halves and
 * synthetic-input / diagnostics entry points.
 *
 * Conventions (unchanged from the reference):
 *   - return value: the HIP runtime's error code cast to unsigned; 0 = success
 *     (panda_interface.cuh:10-16; Rust only tests `!= 0`, unit.rs:55,79)
 *   - handles are `{ void *handle; }` passed BY VALUE; handle = hipStream_t / hipEvent_t / hipMemPool_t
 *   - configuration structs are passed BY VALUE (48 / 48 / 56 bytes)
 *   - panda_msm_execute_* and panda_ntt_execute_* are synchronous on return
 *   - field elements: little-endian u32 limbs in Montgomery form; scalar 32 B; BN254 affine
 *     base 64 B (x||y, identity <=> x == 0); result 96 B X||Y||Z (Jacobian by default,
 *     homogeneous X/Z,Y/Z when PROJECTIVE is requested), identity <=> Z == 0
 *   - unlike the reference (msm_cuda.cuh:155, :554-555) the scalar buffer is never modified and
 *     the device is the caller's current device, not a hard-coded device 0.
 */
#ifndef PANDA_INTERFACE_H
#define PANDA_INTERFACE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ part 1: reference ABI */

typedef enum panda_error /* panda_interface.cuh:10-16 */
{
    panda_success = 0,
    panda_error_invalid_value = 1,
    panda_error_memory_allocation = 2,
    panda_error_not_ready = 600
} panda_error;

typedef struct panda_stream { void *handle; } panda_stream;     /* panda_interface.cuh:18-21 */
typedef struct panda_event { void *handle; } panda_event;       /* panda_interface.cuh:23-26 */
typedef struct panda_mem_pool { void *handle; } panda_mem_pool; /* panda_interface.cuh:28-31 */

typedef enum panda_msm_result_coordinate_type /* panda_interface.cuh:33-37 */
{
    JACOBIAN = 0,
    PROJECTIVE,
} panda_msm_result_coordinate_type;

typedef void (*panda_host_fn)(void *user_data); /* panda_interface.cuh:39 */

/* runtime shim -- panda_interface.cuh:40-68, panda_interface.cu:11-154 */
panda_error panda_get_device_number(int *count);
panda_error panda_get_device(int *device_id);
panda_error panda_set_device(int device_id);
panda_error panda_stream_create(panda_stream *stream, bool blocking_sync);
panda_error panda_stream_wait_event(panda_stream stream, panda_event event);
panda_error panda_stream_sync(panda_stream stream);
panda_error panda_stream_destroy(panda_stream stream);
panda_error panda_launch_host_fn(panda_stream stream, panda_host_fn fn, void *user_data);
panda_error panda_event_create(panda_event *event, bool blocking_sync, bool disable_timing);
panda_error panda_event_record(panda_event event, panda_stream stream);
panda_error panda_event_sync(panda_event event);
panda_error panda_event_query(panda_event event);
panda_error panda_event_destroy(panda_event event);
panda_error panda_mem_get_info(size_t *free, size_t *total);
panda_error panda_malloc(void **ptr, size_t size);
panda_error panda_malloc_host(void **ptr, size_t size);
panda_error panda_free(void *ptr);
panda_error panda_free_host(void *ptr);
panda_error panda_host_register(void *ptr, size_t size);
panda_error panda_host_unregister(void *ptr);
panda_error panda_memcpy(void *dst, const void *src, size_t count);
panda_error panda_memcpy_async(void *dst, const void *src, size_t count, panda_stream stream);
panda_error panda_memset(void *ptr, int value, size_t count);
panda_error panda_memset_async(void *ptr, int value, size_t count, panda_stream stream);
panda_error panda_mem_pool_create(panda_mem_pool *pool, int device_id);
panda_error panda_mem_pool_destroy(panda_mem_pool pool);
panda_error panda_malloc_from_pool_async(void **ptr, size_t size, panda_mem_pool pool, panda_stream stream);
panda_error panda_free_async(void *ptr, panda_stream stream);

typedef struct panda_msm_configuration /* panda_interface.cuh:70-79; Rust MSMConfiguration common.rs:168-185 */
{
    panda_mem_pool mem_pool;
    panda_stream stream;
    void *bases;   /* device: n affine points */
    void *scalars; /* device: n Montgomery-form Fr; read-only here */
    void *results; /* device or pinned host: 3 field elements */
    unsigned log_scalars_count;
    panda_msm_result_coordinate_type msm_result_coordinate_type;
} panda_msm_configuration;
typedef panda_msm_configuration msm_configuration;

panda_error panda_msm_setup_bn254(void);                                               /* panda_interface.cu:152-155 */
panda_error panda_msm_execute_bn254(const panda_msm_configuration exec_cfg);           /* panda_interface.cu:157-160 -> msm_cuda.cuh:551-784 */
panda_error panda_msm_execute_bn254_host(const panda_msm_configuration exec_cfg);      /* panda_interface.cu:162-165 -> msm_host.cuh:267-383; all pointers host */
panda_error panda_msm_tear_down(void);                                                 /* panda_interface.cu:167-170 */

typedef struct panda_ntt_configuration /* panda_interface.cuh:86-94; Rust NTTConfiguration common.rs:187-196 */
{
    panda_mem_pool mem_pool;
    panda_stream stream;
    void *d_src;
    void *d_dst;
    unsigned log_n;
    void *flag; /* host unsigned*: 0 -> result in d_src, 1 -> result in d_dst (fft.cu:211, unit.rs:521-532) */
} panda_ntt_configuration;
typedef panda_ntt_configuration ntt_configuration;

typedef struct panda_ntt_configuration_v1 /* panda_interface.cuh:96-105; Rust NttconfigurationV1 common.rs:198-208 */
{
    panda_mem_pool mem_pool;
    panda_stream stream;
    void *d_src;
    void *d_dst;
    void *d_omega; /* HOST pointer to one Montgomery-form Fr: the primitive 2^log_n-th root (unit.rs:501-511) */
    unsigned log_n;
    void *flag;
} panda_ntt_configuration_v1;
typedef panda_ntt_configuration_v1 ntt_configuration_v1;

panda_error panda_ntt_setup_bn254(void *input_omega);                                  /* panda_interface.cu:172-176 -> fft.cu:225-229 */
panda_error panda_ntt_execute_bn254(panda_ntt_configuration exec_cfg);                 /* panda_interface.cu:178-181 -> fft.cu:231-242 */
panda_error panda_ntt_tear_down(void);                                                 /* panda_interface.cu:188-191 */
panda_error panda_ntt_execute_bn254_v1(const panda_ntt_configuration_v1 exec_cfg);     /* panda_interface.cu:183-186 -> fft.cu:244-260 */

/* ------------------------------------------- part 2: declared by Rust, undefined in the reference */

panda_error panda_stream_synchronize(panda_stream stream); /* binding.rs:14; used by PandaStream::sync, common.rs:71-76 */
panda_error panda_stream_query(panda_stream stream);       /* binding.rs:16 (also declared in panda_interface.cuh:46) */
panda_error panda_device_enable_peer_access(int device_id);  /* binding.rs:56 */
panda_error panda_device_disable_peer_access(int device_id); /* binding.rs:54 */

/* ------------------------------------------------------------------ part 3: additive entry points */

/* BLS12-377 G1 MSM: bases 96 B affine, scalars 32 B, result 144 B (README.md:36 roadmap; BASELINE config 5) */
panda_error panda_msm_setup_bls12_377(void);
panda_error panda_msm_execute_bls12_377(const panda_msm_configuration exec_cfg);
panda_error panda_msm_execute_bls12_377_host(const panda_msm_configuration exec_cfg);
/* BLS12-381 G1 (the reference names the curve, curve.cuh:12, but carries no parameters for it): bases 96 B, result 144 B, curve id 2 */
panda_error panda_msm_setup_bls12_381(void);
panda_error panda_msm_execute_bls12_381(const panda_msm_configuration exec_cfg);
panda_error panda_msm_execute_bls12_381_host(const panda_msm_configuration exec_cfg);

/* BN254 G2 (no counterpart in the reference): the twist y^2 = x^3 + 3/(9+u) over Fq2 = Fq[u]/(u^2+1), scalars as for G1.
 * An Fq2 element is c0 || c1 (2 x 32 B, Montgomery form); affine base x || y = 128 B (identity <=> x == 0), result X || Y || Z = 192 B.
 * Curve id 3 wherever a curve id is taken (register / precompute / execute_from_host / gen_bases / debug_curve_op). */
panda_error panda_msm_setup_bn254_g2(void);
panda_error panda_msm_execute_bn254_g2(const panda_msm_configuration exec_cfg);
panda_error panda_msm_execute_bn254_g2_host(const panda_msm_configuration exec_cfg);
panda_error panda_msm_combine_bn254_g2(const void *partials, unsigned count, panda_msm_result_coordinate_type out_type, void *result);

/* Cached bases (README.md "Supports cached bases and scalars"; init_msm, wrapper.rs:122-152): registering a device buffer
 * of 2^log_n affine bases lets the library keep its radix-converted copy between calls instead of re-deriving it in every
 * panda_msm_execute_*; the caller must not modify the buffer until panda_msm_unregister_bases.  curve: 0 BN254, 1 BLS12-377, 2 BLS12-381, 3 BN254 G2. */
panda_error panda_msm_register_bases(unsigned curve, const void *d_bases, unsigned log_n, panda_stream stream);
panda_error panda_msm_unregister_bases(const void *d_bases);
/* Strict staleness check of a registration: recomputes the 64-bit hash of the whole wire buffer (one streaming pass, about 0.2 ms
 * per GiB; synchronises `stream`) and compares it with the hash taken when the buffer was registered.  panda_success: unchanged;
 * panda_error_invalid_value: not registered, or changed -- the registration has then been dropped.  (Every execute additionally
 * compares 64 sampled rows at no cost; that catches wholesale reuse of the address, not a change confined to other rows.) */
panda_error panda_msm_verify_registered(const void *d_bases, panda_stream stream);
/* on != 0: every panda_msm_execute_* against registered bases runs panda_msm_verify_registered's check first */
panda_error panda_msm_set_paranoid(unsigned on);
/* Cached bases with precomputed window tables (the lookup-table idea the reference left as a stub, msm_host.cuh:248-265):
 * besides the converted copy the library keeps 2^lo[k] * P for every window k (W tables of 2^log_n affine rows, built once,
 * about W * 64 B per BN254 point).  Every window then shares one bucket space, which permits windows of up to 23 bits and
 * removes the per-window reductions.  window_bits 0 = built-in policy.  Results are the same group element as without tables.
 * Undone by panda_msm_unregister_bases. */
panda_error panda_msm_precompute_bases(unsigned curve, const void *d_bases, unsigned log_n, unsigned window_bits, panda_stream stream);
/* what is registered for d_bases: number of tables (1 = converted copy only), window bits (0 if none), device bytes held */
panda_error panda_msm_registered_info(const void *d_bases, unsigned *tables, unsigned *window_bits, size_t *bytes);

/* Upload / execute pipeline inside ONE call (the reference's h2d / exec streams, wrapper.rs:260-273 and unit.rs:17-29, run
 * one after the other).  exec_cfg is as for panda_msm_execute_*; exec_cfg.scalars is the caller's DEVICE buffer of n x 32 bytes and
 * h_scalars the HOST source it is filled from (pinned memory lets the copies run beside the kernels).  Against registered bases the
 * scalars are cut into `ranges` (1..8) contiguous point ranges of n/2^(R-1), n/2^(R-1), n/2^(R-2), ..., n/2 points: range r+1 is copied
 * on h2d_stream while range r runs digits -> sort -> accumulate on exec_cfg.stream against its own rows of the registered tables; each
 * range's buckets are added into a running total on the device, which is reduced once.  Unregistered bases, or fewer than 2^16 points
 * in the first range, reduce the number of ranges, down to one copy followed by the ordinary call.  h_scalars == NULL skips the copies
 * and runs the same schedule on resident scalars.  Synchronous on return like panda_msm_execute_*; same group element.  curve: 0 .. 3
 * (BN254, BLS12-377, BLS12-381, BN254 G2).  Experiment switch: ranges = 0x100 | R with h_scalars == NULL runs R (a power of two) EQUAL ranges
 * one after the other on the caller's stream (the table-footprint measurement of profiles/r05_accumulate_table_footprint.txt). */
panda_error panda_msm_execute_from_host(unsigned curve, const panda_msm_configuration exec_cfg, const void *h_scalars, unsigned ranges, panda_stream h2d_stream);

/* Window size override for experiments: 0 = built-in policy (replaces get_window_bits_count, msm_cuda.cuh:21-45) */
panda_error panda_msm_set_window_bits(unsigned window_bits);
/* what the plain path (no precomputed tables) would run a 2^log_n-point MSM of `curve` with: widest window and number of windows */
panda_error panda_msm_plain_window_plan(unsigned curve, unsigned log_n, unsigned *window_bits, unsigned *windows);
/* sorted entries per thread of the bucket-accumulation kernel, for experiments: 0 = built-in policy (rounded up to a multiple of 4) */
panda_error panda_msm_set_chunk_entries(unsigned entries);
/* With precomputed tables, levels 2 and 3 of the bucket sort for all but the first front_of_128 / 128 of the bucket space run on a second
 * stream of the calling host thread, beside the accumulation of that front part; the rest is then accumulated on that second stream.
 * workgroups_per_cu = 64: the front's accumulation is an ordinary grid; less: that many workgroups per CU whose waves draw their chunks
 * from a counter; 0 = built-in (8 for BN254).  front_of_128: 1 .. 127, 0 = one stream, one phase after the other, 0xffffffff = built-in
 * policy (default: off -- every schedule measured is slower on MI355X, profiles/r05_overlap_sort_accumulate.txt).  Same group element. */
panda_error panda_msm_set_overlap(unsigned front_of_128, unsigned workgroups_per_cu);
/* Experiments on the bucket-accumulation kernel of the 9-limb base fields (BN254): 0 = the built-in choice, 1 = five waves per SIMD with
 * the next entry's table row staged in LDS (global_load_lds) instead of registers, 2 = four waves per SIMD with the staged row, 3 = (every
 * curve) the rows of a wave fetched four lanes to a row into LDS (profiles/r05_accumulate_table_footprint.txt), 4 = (every curve) a
 * chunk's sorted words in whole 64-byte sectors through LDS -- built in for the 9-limb fields from round 6 on --, 5 = never (sixteen-byte
 * global loads, as rounds 2-5 did; profiles/r06_accumulate_fetch_and_row_loads.txt).  Same group element. */
panda_error panda_msm_set_accumulate_variant(unsigned variant);
/* Level-3 merge of the bucket sort (tabled calls): 0 = built-in (the cells of the lower half of the bucket space, which hold up to twice
 * the mean when the plan mixes two window widths, go through the variant that reads up to 32 k entries per cell once, those of the upper
 * half through 256-thread workgroups), 1 = every cell through the wide variant, 2 = neither variant, 3 = every cell through the 256-thread
 * variant (profiles/r05_k3_merge_cells.txt).  Same group element. */
panda_error panda_msm_set_wide_merge(unsigned mode);
/* deprecated no-op kept so that code linked against the round-3 interface still loads (the bucket reduction has no groups any more) */
panda_error panda_msm_set_reduce_group(unsigned log_group);
/* 1 (built in): a small kernel behind the sort hands every chunk of the bucket-accumulation kernel the bucket its first entry falls in;
 * 0: every thread of that kernel finds it by binary search over the bucket offsets, as rounds 1-5 did (A/B measurements). */
panda_error panda_msm_set_chunk_first(unsigned on);
/* Which device timers a call records (an event between two kernels keeps the GPU idle for about 6 us): 0 = none (default),
 * 1 = the call's total + the bucket-accumulation kernel, 2 = every phase.  Phases that were not timed read 0 in panda_msm_last_phase_ms. */
panda_error panda_msm_set_phase_timing(unsigned level);
/* per-phase device times of the last MSM on this thread, milliseconds; names via panda_msm_phase_name */
#define PANDA_MSM_PHASES 8
panda_error panda_msm_last_phase_ms(float *ms /* PANDA_MSM_PHASES floats */);
const char *panda_msm_phase_name(unsigned phase);

/* Inverse transform: runs the forward passes with omega^-1 and fuses the n^-1 scaling into the last pass.
 * d_omega is the FORWARD root (host pointer), as for _v1. */
panda_error panda_ntt_execute_bn254_inverse(const panda_ntt_configuration_v1 exec_cfg);
/* device time of the passes of the last panda_ntt_execute_* on this host thread (HIP events on the launch stream), milliseconds */
panda_error panda_ntt_last_device_ms(float *ms);
/* The passes a natural-order transform of 2^log_n points runs: their number (the reference's loop, fft.cu:171-216, takes eight bits per
 * pass; here 2^17/2^18 and 2^25..2^27 run one pass less behind radix-512 passes) and, if radix_bits != NULL, the bits of each (four
 * entries, zero-padded).  *flag of the execute calls is passes & 1.  The bit-reversed orderings run the same number of passes (radix-512
 * passes last for a bit-reversed input) except at 2^18 and 2^27, where they keep the eight-bit plan. */
panda_error panda_ntt_pass_plan(unsigned log_n, unsigned *passes, unsigned *radix_bits);
/* Streamed inter-pass table: the second boundary of a three-pass transform (2^17 .. 2^26 points) multiplies every element by ONE entry of
 * a table over the whole twiddle index range -- 32 bytes per element of the transform, built once per root, size and direction, kept in
 * the calling thread's twiddle cache (two slots) until panda_ntt_tear_down -- instead of by two entries of 2^16-entry tables.
 *   mode 1 (the default policy): transforms of 2^17 .. 2^24 points; the table is as large as the data, at most 512 MiB per slot, so one host
 *          thread that alternates forward and inverse 2^24-point transforms holds 1 GiB of tables (-6 % at 2^20, -4 % at 2^22, -1.4 % at 2^24)
 *   mode 2: also 2^25 / 2^26 points (1 GiB / 2 GiB per slot, up to 4 GiB per thread; -1.7 % / -0.9 %) -- opt-in
 *   mode 0: off (two small tables, as before round 5);  0xffffffff: back to the built-in policy;  3: test hook (allocation treated as failed)
 * A table that does not fit in free HBM is skipped for that size and device from then on (per host thread); the call still succeeds. */
panda_error panda_ntt_set_streamed_tables(unsigned mode);
/* whole-transform twiddle-table sets built so far by the calling host thread: a repeated transform must not add to it (cache hit) */
panda_error panda_ntt_table_builds(uint64_t *count);
/* Clock stamps (measurement only; off by default).  With panda_set_clock_stamps(1) an MSM brackets the k_accumulate launch of its last
 * range, and a whole NTT its passes, with a marker kernel in which one wave per CU stores s_memtime (shader cycles) and s_memrealtime
 * (100 MHz); stamps are only compared within one CU (the cycle counter is not chip-wide).  panda_*_last_clock fills PANDA_CLOCK_WORDS u64:
 *   [0] shader cycles of the XCD that showed the fewest -- the slowest-clocked one, which the launch waits for (the faster XCDs idle at
 *       the end of a launch and show more): the cycles the CODE needed
 *   [1] 10 ns ticks (median over the CUs)      [2] XCDs stamped on both sides      [3] mean of the XCDs' cycles: what rocprofv3's
 *       GRBM_GUI_ACTIVE / 8 shows for the same launch; [3] / [1] x 100 MHz is the clock the BOX held     [4..11] cycles per XCD
 * A record with cycles and MHz tells a slower kernel from a slower device.  panda_clock_stamp enqueues one marker on `stream` into a
 * block of PANDA_CLOCK_STAMP_BYTES the device can write (clear it first); panda_clock_delta reduces two blocks (host copies) the same way. */
#define PANDA_CLOCK_WORDS 12
#define PANDA_CLOCK_STAMP_BYTES 32768
panda_error panda_set_clock_stamps(unsigned on);
panda_error panda_msm_last_clock(uint64_t *out /* PANDA_CLOCK_WORDS */);
panda_error panda_ntt_last_clock(uint64_t *out /* PANDA_CLOCK_WORDS */);
panda_error panda_clock_stamp(panda_stream stream, void *block /* PANDA_CLOCK_STAMP_BYTES */);
panda_error panda_clock_delta(const void *before, const void *after, uint64_t *out /* PANDA_CLOCK_WORDS */);
/* Bit-reversed orderings (SURVEY 8f-4 "bit-reversed NTT variants"): the forward transform with y[k] stored at bitrev(k), and the inverse
 * (n^-1 fused) of a buffer in that order back to natural-order coefficients.  Chaining them skips two permutations. */
panda_error panda_ntt_execute_bn254_bitrev_out(const panda_ntt_configuration_v1 exec_cfg);
panda_error panda_ntt_execute_bn254_inverse_bitrev_in(const panda_ntt_configuration_v1 exec_cfg);
/* Coset transforms (additive): forward y[k] = sum_j x[j] g^j w^(jk), inverse x[j] = g^-j n^-1 sum_k y[k] w^(-jk); `shift` is a HOST pointer
 * to g in Montgomery form (32 bytes, non-zero), d_omega the forward root as for _v1.  In place on d_src/d_dst with the usual flag protocol. */
panda_error panda_ntt_execute_bn254_coset(const panda_ntt_configuration_v1 exec_cfg, const void *shift);
panda_error panda_ntt_execute_bn254_coset_inverse(const panda_ntt_configuration_v1 exec_cfg, const void *shift);
/* The same transforms over the BLS12-377 scalar field (README.md:36: "easy to encapsulate ... BLS12-377 later") */
panda_error panda_ntt_execute_bls12_377_v1(const panda_ntt_configuration_v1 exec_cfg);
panda_error panda_ntt_execute_bls12_377_inverse(const panda_ntt_configuration_v1 exec_cfg);
/* ... in every variant the BN254 field has (north_star: "NTT butterfly over BN254/BLS12-377"; field parameters:
 * src/cuda/core/curve/bls12_377/paramter.cuh:130-181): bit-reversed orderings and coset transforms, semantics as for the _bn254_ entry points */
panda_error panda_ntt_execute_bls12_377_bitrev_out(const panda_ntt_configuration_v1 exec_cfg);
panda_error panda_ntt_execute_bls12_377_inverse_bitrev_in(const panda_ntt_configuration_v1 exec_cfg);
panda_error panda_ntt_execute_bls12_377_coset(const panda_ntt_configuration_v1 exec_cfg, const void *shift);
panda_error panda_ntt_execute_bls12_377_coset_inverse(const panda_ntt_configuration_v1 exec_cfg, const void *shift);
panda_error panda_ntt_execute_bls12_381_v1(const panda_ntt_configuration_v1 exec_cfg);
panda_error panda_ntt_execute_bls12_381_inverse(const panda_ntt_configuration_v1 exec_cfg);
panda_error panda_ntt_execute_bls12_381_bitrev_out(const panda_ntt_configuration_v1 exec_cfg);
panda_error panda_ntt_execute_bls12_381_inverse_bitrev_in(const panda_ntt_configuration_v1 exec_cfg);
panda_error panda_ntt_execute_bls12_381_coset(const panda_ntt_configuration_v1 exec_cfg, const void *shift);
panda_error panda_ntt_execute_bls12_381_coset_inverse(const panda_ntt_configuration_v1 exec_cfg, const void *shift);

/* Multi-GPU, one process per GPU.  The exchange itself is the caller's (RCCL through
 * torch.distributed or ncclAllGather): these are the per-rank halves either side of it.
 *   MSM  : each rank runs panda_msm_execute_* on its base range -> one partial (96/144 B);
 *          after the all-gather of partials every rank (or rank 0) calls panda_msm_combine_*.
 *   NTT  : four-step over `ranks` slabs; see DESIGN.md section "multi-GPU". */
panda_error panda_msm_combine_bn254(const void *partials /* host or device, count x 96 B Jacobian */, unsigned count,
                                    panda_msm_result_coordinate_type out_type, void *result /* host, 96 B */);
panda_error panda_msm_combine_bls12_377(const void *partials, unsigned count, panda_msm_result_coordinate_type out_type, void *result);
panda_error panda_msm_combine_bls12_381(const void *partials, unsigned count, panda_msm_result_coordinate_type out_type, void *result);

typedef struct panda_ntt_slab_configuration
{
    panda_stream stream;
    void *d_slab;    /* device: this rank's rows, (n / ranks) elements */
    void *d_scratch; /* device: same size */
    void *omega;     /* HOST pointer: primitive n-th root, Montgomery form */
    unsigned log_n;  /* global size */
    unsigned log_ranks;
    unsigned rank;
    void *flag;      /* host unsigned*: which of d_slab / d_scratch holds the step's output */
} panda_ntt_slab_configuration;
/* step 1: local column transforms + inter-slab twiddle; step 2 (after the all-to-all): local row transforms */
panda_error panda_ntt_slab_step1_bn254(const panda_ntt_slab_configuration cfg);
panda_error panda_ntt_slab_step2_bn254(const panda_ntt_slab_configuration cfg);
/* the same steps, enqueued on cfg.stream without waiting (flag is valid on return): step 1 -> all-to-all -> step 2 on one stream needs one
 * synchronisation at the end */
panda_error panda_ntt_slab_step1_bn254_enqueue(const panda_ntt_slab_configuration cfg);
panda_error panda_ntt_slab_step2_bn254_enqueue(const panda_ntt_slab_configuration cfg);
/* The inverse of the sharded transform, from the forward transform's output layout back to its input layout (n^-1 included), by the
 * mirrored steps: inverse_step1 (size-G inverse transforms) -> the same all-to-all -> inverse_step2 (twiddle w^(-r k2) / n, then the local
 * inverse transform).  cfg.omega is the FORWARD root.  Enqueued without waiting, flag valid on return. */
panda_error panda_ntt_slab_inverse_step1_bn254_enqueue(const panda_ntt_slab_configuration cfg);
panda_error panda_ntt_slab_inverse_step2_bn254_enqueue(const panda_ntt_slab_configuration cfg);
/* the four enqueued halves over the BLS12-377 scalar field */
panda_error panda_ntt_slab_step1_bls12_377_enqueue(const panda_ntt_slab_configuration cfg);
panda_error panda_ntt_slab_step2_bls12_377_enqueue(const panda_ntt_slab_configuration cfg);
panda_error panda_ntt_slab_inverse_step1_bls12_377_enqueue(const panda_ntt_slab_configuration cfg);
panda_error panda_ntt_slab_inverse_step2_bls12_377_enqueue(const panda_ntt_slab_configuration cfg);
/* ... and over the BLS12-381 scalar field */
panda_error panda_ntt_slab_step1_bls12_381_enqueue(const panda_ntt_slab_configuration cfg);
panda_error panda_ntt_slab_step2_bls12_381_enqueue(const panda_ntt_slab_configuration cfg);
panda_error panda_ntt_slab_inverse_step1_bls12_381_enqueue(const panda_ntt_slab_configuration cfg);
panda_error panda_ntt_slab_inverse_step2_bls12_381_enqueue(const panda_ntt_slab_configuration cfg);

/* Multi-GPU, ONE process (no reference counterpart: msm_cuda.cuh:554-555 pins device 0, wrapper.rs:38 opens one
 * device, binding.rs:54-56 only declares the peer-access symbols).  A panda_multi_gpu owns one host thread, one stream and -- with
 * PANDA_MULTI_RCCL -- one RCCL communicator per device (ncclCommInitAll); the whole sharded operation is one call:
 *   panda_msm_execute_*_multi   cfgs[d] is an ordinary panda_msm_configuration whose buffers live on devices[d] and describe that
 *                               device's base-point range (stream.handle == NULL: the handle's own stream of that device).  Every device
 *                               runs the single-GPU pipeline on its worker thread and leaves its JACOBIAN partial in cfgs[d].results;
 *                               one ncclAllGather moves the partials, the sum in cfgs[0]'s coordinate type goes to `result` (HOST, 96 /
 *                               144 bytes).  Synchronous on return.
 *   panda_ntt_execute_*_multi   cfgs[d] is the slab configuration of rank d (rank == d, log_ranks == log2(n_dev), a power of two):
 *                               step 1 on every device, ONE grouped ncclSend / ncclRecv all-to-all (chunk q of rank d to rank q), step 2
 *                               on every device, one synchronisation at the end.  *cfgs[d].flag (HOST) = 1 if rank d's output sits in
 *                               its d_scratch, 0 if in its d_slab.  Layouts as for the panda_ntt_slab_* halves above.
 * PANDA_MULTI_LOOPBACK replaces RCCL by device-to-device copies and lets one device play several ranks (tests on a one-GPU box; also
 * what a caller without xGMI peers gets).  The worker threads keep the library's per-thread scratch and twiddle caches alive between
 * calls; per-device setup (allocation, panda_msm_precompute_bases) is done by the caller under panda_set_device(devices[d]). */
typedef struct panda_multi_gpu { void *handle; } panda_multi_gpu;
#define PANDA_MULTI_RCCL 0u
#define PANDA_MULTI_LOOPBACK 1u
panda_error panda_multi_gpu_create(panda_multi_gpu *out, const int *devices, unsigned n_dev, unsigned transport);
panda_error panda_multi_gpu_destroy(panda_multi_gpu mg);
panda_error panda_multi_gpu_device_count(panda_multi_gpu mg, unsigned *n_dev);
panda_error panda_msm_execute_bn254_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs /* n_dev */, void *result /* host */);
panda_error panda_msm_execute_bls12_377_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, void *result);
panda_error panda_msm_execute_bls12_381_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, void *result /* 144 B */);
panda_error panda_msm_execute_bn254_g2_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, void *result /* 192 B */);
/* The same with the scalars starting on the HOST ("scalars H2D'd per shard"; replaces the staging of unit.rs:103-188, which
 * uploads before it executes): h_scalars[d] is rank d's host source (pinned memory lets the copies run beside the kernels, pageable
 * memory works), cfgs[d].scalars the device buffer it lands in.  Every device's worker runs panda_msm_execute_from_host on its shard --
 * `ranges` point ranges, upload of range r+1 beside the kernels of range r, on the device's own copy stream and PCIe link -- so the
 * n_dev uploads run side by side; then the one ncclAllGather and the combine as above. */
panda_error panda_msm_execute_bn254_from_host_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs /* n_dev */, const void *const *h_scalars /* n_dev */,
                                                    unsigned ranges, void *result /* host */);
panda_error panda_msm_execute_bls12_377_from_host_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, const void *const *h_scalars, unsigned ranges,
                                                        void *result);
panda_error panda_msm_execute_bls12_381_from_host_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, const void *const *h_scalars, unsigned ranges,
                                                        void *result);
panda_error panda_msm_execute_bn254_g2_from_host_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, const void *const *h_scalars, unsigned ranges,
                                                       void *result);
panda_error panda_ntt_execute_bn254_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs /* n_dev */);
panda_error panda_ntt_execute_bn254_inverse_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs);
/* A batch of `count` sharded transforms (same size and root; cfgs[t * n_dev + d] = rank d of transform t, every transform with its own slab and
 * scratch) pipelined over two streams per device: the all-to-all of transform t runs while step 1 of transform t + 1 and step 2 of transform
 * t - 1 compute.  Layouts, flags and results as `count` separate panda_ntt_execute_bn254[_inverse]_multi calls; one synchronisation at the end. */
panda_error panda_ntt_execute_bn254_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs /* count x n_dev */, unsigned count);
panda_error panda_ntt_execute_bn254_inverse_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs, unsigned count);
/* the sharded transforms over the BLS12-377 scalar field */
panda_error panda_ntt_execute_bls12_377_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs /* n_dev */);
panda_error panda_ntt_execute_bls12_377_inverse_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs);
panda_error panda_ntt_execute_bls12_377_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs /* count x n_dev */, unsigned count);
panda_error panda_ntt_execute_bls12_377_inverse_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs, unsigned count);
panda_error panda_ntt_execute_bls12_381_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs /* n_dev */);
panda_error panda_ntt_execute_bls12_381_inverse_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs);
panda_error panda_ntt_execute_bls12_381_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs /* count x n_dev */, unsigned count);
panda_error panda_ntt_execute_bls12_381_inverse_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs, unsigned count);
/* per-phase device times of rank's last MSM inside a *_multi call (the workers' panda_msm_last_phase_ms) */
panda_error panda_multi_gpu_last_phase_ms(panda_multi_gpu mg, unsigned rank, float *ms /* PANDA_MSM_PHASES floats */);

/* Synthetic inputs generated on the device; curve: 0 = BN254, 1 = BLS12-377, 2 = BLS12-381, 3 = BN254 G2 */
panda_error panda_gen_scalars(unsigned curve, uint64_t seed, uint64_t first, uint64_t n, void *d_out, panda_stream stream);
panda_error panda_gen_bases(unsigned curve, uint64_t seed, uint64_t first, uint64_t n, void *d_out, panda_stream stream);

/* Element-wise diagnostics used by the parity tests: field id 0..5 = BN254 Fq, BN254 Fr, BLS12-377 Fq, BLS12-377 Fr, BLS12-381 Fq, BLS12-381 Fr;
 * op 0..6 = add, sub, mul, sqr, to_montgomery, from_montgomery, inverse (Montgomery in/out, 0 -> 0; field.cuh:925-972).  Device pointers. */
panda_error panda_debug_field_op(unsigned field_id, unsigned op, void *d_r, const void *d_a, const void *d_b, size_t n, panda_stream stream);
/* op 0 = Jacobian + affine (madd), 1 = Jacobian + Jacobian, 2 = double, 3 / 4 = the four-lane spellings of 1 / 2 that the MSM's
 * fix-up and bucket-reduction trees run (csrc/curve29_quad.h); Jacobian in/out */
panda_error panda_debug_curve_op(unsigned curve, unsigned op, void *d_r, const void *d_a, const void *d_b, size_t n, panda_stream stream);

const char *panda_version(void);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* PANDA_INTERFACE_H */
