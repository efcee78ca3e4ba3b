// msm_impl.h -- Pippenger multi-scalar multiplication for gfx950: kernels and the per-curve driver templates.
// Instantiated once per curve (msm_bn254.hip, msm_bls377.hip, msm_bls381.hip) so that the heavy translation units build in parallel;
// msm.hip holds the curve-independent host side (arena, cached-bases registry, policies, C ABI).
//
// Drop-in for the reference's GPU path behind panda_msm_execute_bn254
// (src/cuda/core/unit/msm/msm_cuda.cuh:551-784), redesigned rather than translated:
//
//   reference (msm_cuda.cuh)                           here
//   ------------------------------------------------   -----------------------------------------------
//   scalars de-Montgomeryed IN PLACE (:148-157)        digits kernel reads scalars, never writes them
//   unsigned c-bit digits, 2^c - 1 buckets/window      signed digits, 2^(c-1) buckets/window (base negation is free)
//   one thread per bucket walks its list (:373-409)    flat chunks of K sorted entries per thread: every thread does
//     -> collapses on skewed scalars                     exactly K mixed adds whatever the bucket sizes; bucket pieces
//                                                        cut by a chunk boundary are merged by a fix-up kernel
//   each bucket weighted by c doublings + adds         segmented running sums (2 adds per bucket), the weighted sum of the
//     (:411-420, ~240 mulmods per bucket)                group sums bit by bit with plain tree sums (no scalar multiplications)
//   Jacobian madd 7M+4S on 8x32-bit PTX carry chains   XYZZ madd 8M+2S on 9x29-bit limbs / v_mad_u64_u32 (fe29.h)
//   9 cudaDeviceSynchronize per call (:611-755)        one stream, one synchronisation before the host Horner
//   scratch cudaMallocAsync'd and freed per call       per-thread arena kept between calls
//
// Pipeline (all on cfg.stream):
//   k_convert_bases   wire affine -> internal Montgomery radix, 64 B/point (96 B for the BLS curves); skipped for registered bases
//   msm_sort.hip      scalars -> signed window digits -> per-bucket lists of base indices (two or three LDS-staged sort levels)
//   k_accumulate      flat chunks of K entries: acc += +/- base   (the hot kernel)
//   k_fixup           merge bucket pieces that straddle chunks
//   k_sum_lines / k_weighted_slots / k_slot_total
//                     weighted sum of the group sums -> one point per list
//   host              Horner over the W window sums (as the reference does, msm_cuda.cuh:738-743) -- none with precomputed
//                     tables, whose single list already carries the 2^lo[k] factors -- and output conversion
#pragma once
#include <algorithm>
#include <vector>

#include "curve29_quad.h"
#include "msm_sort.h"
#include "panda_internal.h"

namespace panda {

// what the cached-bases registry (msm.hip) holds for a registered device buffer
struct MsmRegistration {
    const void *wire; // the caller's device pointer (the key)
    void *converted;  // n rows (plain) or plan.W tables of n rows (tabled)
    unsigned log_n, curve;
    int device;
    bool tabled;
    WindowPlan plan;
    size_t bytes;            // converted rows / tables (what panda_msm_registered_info reports)
    const uint32_t *samples; // REG_SAMPLES rows of the wire buffer as registered, kept behind the tables (stale-address check)
    uint64_t hash;           // of the whole wire buffer as registered (msm_hash_wire): the strict check, on demand
};
// order-independent 64-bit hash of `bytes` (a multiple of 16) of device memory, on `s`; synchronises
hipError_t msm_hash_wire(const void *d_buf, size_t bytes, hipStream_t s, uint64_t *hash);

struct MsmTuning {
    unsigned window_bits;  // plain-mode window width (already resolved by the policy)
    unsigned chunk;        // sorted entries per k_accumulate thread, 0 = built-in
    unsigned timing;       // 0: no device timers at all; 1: the call's total + k_accumulate; 2: every phase (an event between two kernels costs ~6 us of idle GPU)
    unsigned overlap_front; // with tables: size of the front part of the bucket space, in 1/128, whose accumulation runs beside the sort of the rest (0 = no overlap)
    unsigned overlap_wgs;   // workgroups per CU of that accumulation (6 = three waves per SIMD); 0 = the curve's default
    unsigned chunk_first;   // 1 (built in): k_chunk_first hands every chunk of k_accumulate its first bucket; 0: every thread searches the offsets for it (round 1-5)
    unsigned acc_variant;   // experiments on k_accumulate: 0 = the built-in kernel, 1 = five waves per SIMD with the next row staged in LDS, 2 = four waves with it (9-limb fields), 3 = rows fetched four lanes to a row (k_accumulate_shared), 4 = the sorted words in 64-byte sectors through LDS for every field (built in for the 9-limb fields), 5 = never (sixteen-byte global loads, rounds 2-5)
};

// Point-range pipeline inside one call (the reference's three streams, wrapper.rs:260-273, unit.rs:17-29, serialise
// upload and execution): the scalars are cut into contiguous point ranges; range r+1 crosses PCIe on `h2d` while range r runs
// digits -> sort -> accumulate against its own rows of the registered tables; each range's fix-up adds its buckets into the running
// total on the device, which is reduced once.  h_scalars == nullptr runs the same chunked schedule on scalars that are already resident.
struct MsmPipeline {
    const void *h_scalars; // host (ideally pinned) source of the n x 32 B scalars, copied into cfg.scalars range by range
    unsigned ranges;       // number of point ranges R (1 = whole call at once): n/2^(R-1), n/2^(R-1), n/2^(R-2), ..., n/2 points
    hipStream_t h2d;       // stream the copies are issued on
};

// per-curve entry points (defined in msm_bn254.hip / msm_bls377.hip / msm_bls381.hip).  *stale is set when the caller's buffer
// no longer matches the rows sampled at registration; the result written is then meaningless and msm.hip repeats the call
hipError_t msm_execute_bn254(const panda_msm_configuration &cfg, const MsmRegistration *reg, MsmTuning tuning, float *phase_ms, bool *stale,
                              const MsmPipeline *pipe);
hipError_t msm_execute_bls377(const panda_msm_configuration &cfg, const MsmRegistration *reg, MsmTuning tuning, float *phase_ms, bool *stale,
                              const MsmPipeline *pipe);
hipError_t msm_build_registration_bn254(MsmRegistration &r, hipStream_t s);
hipError_t msm_build_registration_bls377(MsmRegistration &r, hipStream_t s);
hipError_t msm_execute_bls381(const panda_msm_configuration &cfg, const MsmRegistration *reg, MsmTuning tuning, float *phase_ms, bool *stale,
                              const MsmPipeline *pipe);
hipError_t msm_build_registration_bls381(MsmRegistration &r, hipStream_t s);
hipError_t msm_execute_bn254_g2(const panda_msm_configuration &cfg, const MsmRegistration *reg, MsmTuning tuning, float *phase_ms, bool *stale,
                                const MsmPipeline *pipe);
hipError_t msm_build_registration_bn254_g2(MsmRegistration &r, hipStream_t s);

// scalar field of a curve id (what the digit extraction and the window plans are keyed on): BN254 G2 shares BN254's
static constexpr inline unsigned msm_scalar_field_of(unsigned curve) { return curve == 3 ? 0u : curve; }

} // namespace panda

#ifdef PANDA_MSM_IMPL // the kernels and templates below are only wanted by the per-curve translation units

using namespace panda29;

namespace {



struct CurveBn254 {
    typedef Bn254Fq Fq;
    typedef Bn254Fr Fr;
    static constexpr unsigned ID = 0; // curve id of the C ABI; also selects the scalar field in msm_sort
};
struct CurveBls377 {
    typedef Bls377Fq Fq;
    typedef Bls377Fr Fr;
    static constexpr unsigned ID = 1;
};
struct CurveBls381 {
    typedef Bls381Fq Fq;
    typedef Bls381Fr Fr;
    static constexpr unsigned ID = 2;
};
// BN254 G2: the twist over Fq2 = Fq[u] / (u^2 + 1); same scalar field, coordinates of 2 x 8 wire words (affine base 128 B, result 192 B)
struct CurveBn254G2 {
    typedef Ext2<Bn254Fq> Fq;
    typedef Bn254Fr Fr;
    static constexpr unsigned ID = 3;
};

// ------------------------------------------------------------------------------- HBM layouts
// XYZZ point: 4*N u32, array of structs (144 B for N = 9, 224 B for N = 14; both multiples of 16).
template <class F>
__device__ __forceinline__ void store_xyzz(u32 *dst, const Xyzz<F> &p)
{
    constexpr int N = F::N;
    u32 tmp[4 * N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        tmp[i] = p.X.l[i];
        tmp[N + i] = p.Y.l[i];
        tmp[2 * N + i] = p.ZZ.l[i];
        tmp[3 * N + i] = p.ZZZ.l[i];
    }
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
#pragma unroll
    for (int i = 0; i < N; i++) d4[i] = make_uint4(tmp[4 * i], tmp[4 * i + 1], tmp[4 * i + 2], tmp[4 * i + 3]);
}

template <class F>
__device__ __forceinline__ void load_xyzz(Xyzz<F> &p, const u32 *src)
{
    constexpr int N = F::N;
    u32 tmp[4 * N];
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint4 v = s4[i];
        tmp[4 * i] = v.x;
        tmp[4 * i + 1] = v.y;
        tmp[4 * i + 2] = v.z;
        tmp[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        p.X.l[i] = tmp[i];
        p.Y.l[i] = tmp[N + i];
        p.ZZ.l[i] = tmp[2 * N + i];
        p.ZZZ.l[i] = tmp[3 * N + i];
    }
}

template <int WORDS>
__device__ __forceinline__ void load_words(u32 *dst, const u32 *src)
{
    static_assert(WORDS % 4 == 0, "vector loads");
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
#pragma unroll
    for (int i = 0; i < WORDS / 4; i++) {
        uint4 v = s4[i];
        dst[4 * i] = v.x;
        dst[4 * i + 1] = v.y;
        dst[4 * i + 2] = v.z;
        dst[4 * i + 3] = v.w;
    }
}

// the same with the non-temporal hint: data that is read exactly once (the table rows k_accumulate gathers) should not push data that is
// read again (the sorted words, four 16-byte loads per 64-byte sector, minutes of cache time apart) out of the L2
template <int WORDS>
__device__ __forceinline__ void load_words_stream(u32 *dst, const u32 *src)
{
    static_assert(WORDS % 4 == 0, "vector loads");
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const v4u *s4 = reinterpret_cast<const v4u *>(src);
#pragma unroll
    for (int i = 0; i < WORDS / 4; i++) {
        const v4u v = __builtin_nontemporal_load(&s4[i]);
        dst[4 * i] = v.x;
        dst[4 * i + 1] = v.y;
        dst[4 * i + 2] = v.z;
        dst[4 * i + 3] = v.w;
    }
}

template <int WORDS>
__device__ __forceinline__ void store_words(u32 *dst, const u32 *src)
{
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
#pragma unroll
    for (int i = 0; i < WORDS / 4; i++) d4[i] = make_uint4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
}

// ------------------------------------------------------------------------------- kernels

// wire affine (Montgomery radix 2^(32L)) -> internal radix 2^(29N), canonical, packed in 2*L words.
// A wire identity (x == 0, affine.cuh:72-75) becomes all zeros.
template <class F>
__global__ void __launch_bounds__(256) k_convert_bases(const u32 *__restrict__ wire, u32 *__restrict__ out, u64 n)
{
    constexpr int L = F::L;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[2 * L], o[2 * L];
    load_words<2 * L>(w, wire + i * 2 * L);
    Fe<F> x, y;
    bool inf = affine_from_wire(x, y, w);
    fe_reduce_once(x);
    fe_reduce_once(y);
    fe_pack(o, x);
    fe_pack(o + L, y);
    if (inf) {
#pragma unroll
        for (int k = 0; k < 2 * L; k++) o[k] = 0;
    }
    store_words<2 * L>(out + i * 2 * L, o);
}

// Group operations of the kernels outside the hot loop.  Over Fq2 (G2) one full addition is ~50 base multiplications; inlined a dozen
// times per kernel it made this translation unit take six minutes to compile, so there the merge / reduction kernels call ONE
// out-of-line copy (G2 is not the path the benchmark times); the prime-field curves keep the inlined code.
template <class F>
__device__ __noinline__ void xyzz_add_outlined(Xyzz<F> &acc, const Xyzz<F> &q)
{
    xyzz_add(acc, q);
}
template <class F>
__device__ __noinline__ void xyzz_dbl_outlined(Xyzz<F> &r, const Xyzz<F> &p)
{
    xyzz_dbl(r, p);
}
template <class F>
__device__ __forceinline__ void xyzz_add_k(Xyzz<F> &acc, const Xyzz<F> &q)
{
    if constexpr (IsExt2<F>::value)
        xyzz_add_outlined(acc, q);
    else
        xyzz_add(acc, q);
}
template <class F>
__device__ __forceinline__ void xyzz_dbl_k(Xyzz<F> &r, const Xyzz<F> &p)
{
    if constexpr (IsExt2<F>::value)
        xyzz_dbl_outlined(r, p);
    else
        xyzz_dbl(r, p);
}

// Four-lane spellings (curve29_quad.h) for the trees behind k_accumulate, where waves run alone on their SIMD and a level of the
// tree costs one addition's latency: 3.4 us instead of 7.3 (profiles/r04_ubench_addlat.txt).  role = lane & 3; operands and result
// are replicated over the quad.
template <class F>
__device__ __noinline__ void xyzz_add_quad_outlined(Xyzz<F> &acc, const Xyzz<F> &q, unsigned role)
{
    xyzz_add_quad(acc, q, role);
}
template <class F>
__device__ __noinline__ void xyzz_dbl_quad_outlined(Xyzz<F> &r, const Xyzz<F> &p, unsigned role)
{
    xyzz_dbl_quad(r, p, role);
}
template <class F>
__device__ __forceinline__ void xyzz_add_q(Xyzz<F> &acc, const Xyzz<F> &q, unsigned role)
{
    if constexpr (IsExt2<F>::value)
        xyzz_add_quad_outlined(acc, q, role);
    else
        xyzz_add_quad(acc, q, role);
}
template <class F>
__device__ __forceinline__ void xyzz_dbl_q(Xyzz<F> &r, const Xyzz<F> &p, unsigned role)
{
    if constexpr (IsExt2<F>::value)
        xyzz_dbl_quad_outlined(r, p, role);
    else
        xyzz_dbl_quad(r, p, role);
}

// Rows of the caller's wire buffer remembered at registration; the digits kernel of every execute compares them (msm_sort.hip,
// check_samples; msm.hip, "Staleness").  Block = REG_SAMPLES x 16 threads, lane l owns words l, l+16, ...
__global__ void __launch_bounds__(panda::REG_SAMPLES * 16) k_take_samples(const u32 *__restrict__ wire, u32 *__restrict__ samples, u64 n, unsigned row_words)
{
    const unsigned t = threadIdx.x >> 4, l = threadIdx.x & 15;
    const u64 row = panda::sample_row(t, n);
    for (unsigned k = l; k < row_words; k += 16) samples[t * row_words + k] = wire[row * row_words + k];
}

// first index in off[0..NB] whose value exceeds pos, minus one: the bucket that owns sorted position pos
__device__ __forceinline__ u32 owner_bucket(const u32 *off, u32 NB, u32 pos)
{
    u32 lo = 0, hi = NB; // invariant: off[lo] <= pos < off[hi]
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (off[mid] <= pos) lo = mid;
        else hi = mid;
    }
    return lo;
}

// The bucket that owns the first entry of every chunk, for all chunks at once: bucket b covers sorted positions [off[b], off[b + 1]) and
// hands itself to the chunks that START there (t K in that range).  One coalesced pass over the offsets and chunks x 4 bytes written,
// instead of a 21-step binary search over 8 MB of offsets in each of k_accumulate's 1.6 M threads (2^24 points: ~2 GB of the launch's
// 16.2 GB of fetches and 21 dependent loads in front of every chunk).  A bucket long enough to start many chunks
// (skewed scalars: all-equal puts n / K chunks into one bucket) is spread over the lanes of its wave.
__global__ void __launch_bounds__(256) k_chunk_first(const u32 *__restrict__ off, u32 *__restrict__ first, unsigned NB, unsigned K, unsigned chunks)
{
    const unsigned w = blockIdx.y;
    const u32 *ow = off + (u64)w * (NB + 1);
    u32 *fw = first + (u64)w * chunks;
    const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
    u32 t0 = 0, t1 = 0;
    if (b < NB) {
        const u32 lo = ow[b], hi = ow[b + 1];
        t0 = (lo + K - 1) / K;
        t1 = min((u32)((hi + K - 1) / K), (u32)chunks);
    }
    const unsigned lane = threadIdx.x & 63u;
    u64 wide = __ballot(t1 > t0 + 4);
    while (wide) {
        const int src = __ffsll((long long)wide) - 1;
        wide &= wide - 1;
        const u32 a = (u32)__shfl((int)t0, src), e = (u32)__shfl((int)t1, src), owner = (u32)__shfl((int)b, src);
        for (u32 t = a + lane; t < e; t += 64) fw[t] = owner;
    }
    if (t1 <= t0 + 4)
        for (u32 t = t0; t < t1; t++) fw[t] = b;
}

// measurement builds only (-DPANDA_ROW_MASK=0x03ffffff: every gather confined to the first 4 GiB of rows; results are then wrong)
#ifndef PANDA_ROW_MASK
#define PANDA_ROW_MASK 0x7fffffffu
#endif
// a converted base as it sits in HBM: 2*L words, all zero for the identity
template <class F>
struct PackedBase {
    u32 w[2 * F::L];
};

template <class F>
__device__ __forceinline__ void fetch_base(PackedBase<F> &b, const u32 *bases, u32 entry)
{
#if defined(PANDA_ROWS_NT)
    load_words_stream<2 * F::L>(b.w, bases + (u64)(entry & PANDA_ROW_MASK) * 2 * F::L);
#else
    load_words<2 * F::L>(b.w, bases + (u64)(entry & PANDA_ROW_MASK) * 2 * F::L);
#endif
}

// RAW: a negated y comes back un-normalised (limbs < 2^31) -- good enough for the one product it feeds in
// xyzz_madd_core, three instructions per limb cheaper; callers that store or square y normalise it first
template <class F, bool RAW = false>
__device__ __forceinline__ void unpack_base(Fe<F> &x, Fe<F> &y, bool &inf, const PackedBase<F> &b, u32 entry)
{
    constexpr int L = F::L;
    u32 nz = 0;
#pragma unroll
    for (int k = 0; k < 2 * L; k++) nz |= b.w[k];
    inf = (nz == 0);
    fe_unpack(x, b.w);
    fe_unpack(y, b.w + L);
    if (entry >> 31) {
        Fe<F> ny;
        if constexpr (RAW)
            fe_neg_raw<F, 1>(ny, y); // y is canonical (< p)
        else
            fe_neg<F, 1>(ny, y);
        y = ny;
    }
}

// The hot kernel.  Thread t of window w owns sorted entries [t*K, t*K + K) of that window and adds the
// bases they name into the accumulators of the buckets they fall in; the first and last bucket of a
// chunk may continue in the neighbouring chunks, those pieces go to `parts` and are merged by k_fixup.
// Replaces aggerate_buckets_groups_kernel's per-bucket list walk (msm_cuda.cuh:373-409).
#ifndef ACC_RAW_Y
#define ACC_RAW_Y false
#endif
// Which chunks of the list a launch owns.  pos == nullptr: all of them (one launch per list).  Otherwise the sort was split
// (SortSplit, msm_sort.h): pos[] are the positions of the level-3 cells in the list, and the launch owns the chunks that END in
// (pos[lo], pos[hi]] -- every entry and every bucket offset such a chunk reads is final once the cells below `hi` are merged, so the
// front of the list is accumulated while the rest is still being sorted.  `last` marks the launch that also takes the list's final,
// possibly short chunk; hi_bucket is the first bucket of cell `hi` (the offsets beyond it may not exist yet).
struct AccPart {
    const u32 *pos;
    unsigned lo, hi, last, hi_bucket;
    u32 *queue; // PERSIST launches: a counter, zero at launch, from which every wave draws its next 64 chunks
};

// LDSROW: the row of the NEXT entry travels from HBM straight into LDS (global_load_lds_dwordx4: lane l's 16-byte pieces land at
// piece * 1 KB + 16 l of the wave's region) instead of sitting packed in 2 L registers for the whole addition it is fetched under; it is
// read back when its turn comes.  That frees 16 (24) registers through the long part of the loop -- what a fifth wave per SIMD needs.
template <class F>
__device__ __forceinline__ void fetch_base_lds(uint4 *lds_wave, const u32 *bases, u32 entry)
{
    const u32 *src = bases + (u64)(entry & PANDA_ROW_MASK) * 2 * F::L;
#pragma unroll
    for (int j = 0; j < 2 * F::L / 4; j++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * j), (__attribute__((address_space(3))) void *)(lds_wave + j * 64), 16, 0, 0);
}
template <class F>
__device__ __forceinline__ void read_base_lds(PackedBase<F> &b, const uint4 *lds_wave, unsigned lane)
{
    // the row was sent to LDS one iteration ago; the compiler's own wait in front of an aliasing LDS read is not placed on every path of
    // a loop-carried transfer (the first build read stale rows), so it is spelled out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 2 * F::L / 4; j++) {
        const uint4 v = lds_wave[j * 64 + lane];
        b.w[4 * j] = v.x;
        b.w[4 * j + 1] = v.y;
        b.w[4 * j + 2] = v.z;
        b.w[4 * j + 3] = v.w;
    }
}

// SWLDS (experiment, panda_msm_set_accumulate_variant(4)): the chunk's sorted words travel in whole 64-byte sectors -- four
// global_load_lds per sixteen entries into one of two 64-byte-per-lane LDS buffers of the wave, a ds_read_b128 per four entries -- instead
// of one 16-byte global load per four entries.  A lane walks its own stretch of the list, so its four loads of one sector are four
// additions (~18 us) apart, and by then the sector has left the L2: FETCH_SIZE counts the sorted words (0.8 GB at 2^24) about four times.
// lds_words: the wave's 2 buffers x 4 pieces x 64 lanes of uint4.
template <class F, bool LDSROW, bool SWLDS = false>
__device__ __forceinline__ void accumulate_chunk(const u32 *__restrict__ bases, const u32 *__restrict__ sw, const u32 *__restrict__ ow, u32 *__restrict__ bw, u32 *__restrict__ pw,
                                                 u32 b, u32 start, u32 end, uint4 *lds_wave, unsigned lane, uint4 *lds_words = nullptr)
{
    constexpr int PW = 4 * F::N;
    u32 next = ow[b + 1];
    bool run_starts_inside = ow[b] >= start; // only the first run of a chunk can have begun in an earlier chunk
    Xyzz<F> acc;
    xyzz_set_identity(acc);

    // Software pipeline: while the addition of entry `pos` runs, the base of entry pos+1 (still packed: 16 registers)
    // and the sorted word of entry pos+2 are in flight, so neither the gather nor the dependent address load is waited
    // for at its point of issue.  The run-boundary work sits BEFORE the gather is issued: its dependent load of the next
    // boundary waits on everything outstanding (the counter is in order), and with the fresh gather among it a wave --
    // which meets a boundary in one iteration out of two -- stalled for an HBM round trip each time (k_accumulate ran
    // 14.5 ms with the rows in HBM against 12.2 ms with L2-resident rows: profiles/r02_accumulate_stalls.txt).
    PackedBase<F> next_base;
    // The sorted words of a chunk are read four at a time: a lane walks its own 512-byte stretch of the list, so a 4-byte load per
    // entry fetched the same 64-byte sector sixteen times over -- long after the L2 had let go of it (FETCH_SIZE: 17.8 GB per launch
    // against 12.9 GB of rows + 0.8 GB of words).
    const bool quads = ((reinterpret_cast<uintptr_t>(sw + start) & 15) == 0); // chunk starts are multiples of K >= 16 words; tiny lists may not be aligned
    // sectors: the chunk starts on 64 bytes (K a multiple of 16 words); sector s of the chunk sits in buffer s & 1
    const bool sectors = SWLDS && ((reinterpret_cast<uintptr_t>(sw + start) & 63) == 0);
    auto sector_to_lds = [&](u32 sector) { // at most 60 bytes past the list's end, inside the arena
        const u32 *src = sw + start + 16 * sector;
#pragma unroll
        for (int q = 0; q < 4; q++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * q),
                                             (__attribute__((address_space(3))) void *)(lds_words + (sector & 1u) * 256 + q * 64), 16, 0, 0);
    };
    uint4 quad = make_uint4(0, 0, 0, 0);
    if constexpr (SWLDS) {
        if (sectors) {
            sector_to_lds(0);
            if (start + 16 < end) sector_to_lds(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing else is in flight yet
            quad = lds_words[lane];
        } else if (quads)
            quad = *reinterpret_cast<const uint4 *>(sw + start);
    } else if (quads)
        quad = *reinterpret_cast<const uint4 *>(sw + start); // at most 12 bytes past the list's end, inside the arena
    u32 cur_entry = quads ? quad.x : sw[start];
    u32 ahead_entry = quads ? quad.y : (start + 1 < end ? sw[start + 1] : 0u);
    if constexpr (LDSROW)
        fetch_base_lds<F>(lds_wave, bases, cur_entry);
    else
        fetch_base<F>(next_base, bases, cur_entry);
    for (u32 pos = start; pos < end; pos++) {
        Fe<F> cx, cy;
        bool cinf;
        const u32 entry = cur_entry;
        if constexpr (LDSROW) read_base_lds<F>(next_base, lds_wave, lane);
        unpack_base<F, ACC_RAW_Y && RawOperandOk<F>::value>(cx, cy, cinf, next_base, entry);
        const bool boundary = pos >= next;
        if (boundary) { // the run of bucket b ends here; this entry opens the next run, so it simply becomes the accumulator
            store_xyzz<F>(run_starts_inside ? bw + (u64)b * PW : pw, acc); // its end (== pos) is inside the chunk by construction
            run_starts_inside = true;
            do {
                b++;
                next = ow[b + 1];
            } while (next <= pos);
            if (cinf)
                xyzz_set_identity(acc);
            else {
                Fe<F> ty;
                fe_norm(ty, cy);
                xyzz_from_affine(acc, cx, ty);
            }
        }
        if (pos + 1 < end) {
            cur_entry = ahead_entry;
            if constexpr (LDSROW)
                fetch_base_lds<F>(lds_wave, bases, cur_entry);
            else
                fetch_base<F>(next_base, bases, cur_entry);
            if (pos + 2 < end) {
                if (quads) {
                    const u32 idx = pos + 2 - start;
                    if ((idx & 3u) == 0) {
                        if (SWLDS && sectors) {
                            // the sector was sent to LDS sixteen entries ago (or at the start), i.e. before the row gather that the top of this
                            // iteration waited for: the counter is in order, so it has landed.  Entering sector s frees the other buffer (sector
                            // s - 1 has been read to its end) for sector s + 1.
                            if ((idx & 15u) == 0 && start + idx + 16 < end) sector_to_lds((idx >> 4) + 1);
                            quad = lds_words[((idx >> 4) & 1u) * 256 + ((idx >> 2) & 3u) * 64 + lane];
                        } else
                            quad = *reinterpret_cast<const uint4 *>(sw + pos + 2);
                    }
                    ahead_entry = (idx & 2u) ? ((idx & 1u) ? quad.w : quad.z) : ((idx & 1u) ? quad.y : quad.x);
                } else
                    ahead_entry = sw[pos + 2];
            }
        }
        if (boundary) continue;
        if (cinf) continue;
        if (xyzz_is_identity(acc)) {
            Fe<F> ty;
            fe_norm(ty, cy); // a raw negated y must not be stored
            xyzz_from_affine(acc, cx, ty);
            continue;
        }
        const int rare = xyzz_madd_core(acc, cx, cy);
        if (rare) { // same x as the accumulator: reload the base instead of keeping it live through the common path
            if (rare == 1) {
                PackedBase<F> again;
                fetch_base<F>(again, bases, entry);
                unpack_base<F>(cx, cy, cinf, again, entry);
                xyzz_dbl_affine(acc, cx, cy);
            } else
                xyzz_set_identity(acc);
        }
    }
    // last run: complete only if the bucket both starts and ends inside the chunk
    const bool starts_inside = run_starts_inside;
    const bool ends_inside = next <= end;
    u32 *dst = (starts_inside && ends_inside) ? bw + (u64)b * PW : (starts_inside ? pw + PW : pw);
    store_xyzz<F>(dst, acc);
}

// PERSIST: the grid is a fixed number of workgroups (a few per CU), all resident at once, whose waves draw their chunks -- 64
// consecutive ones at a time -- from a counter (part.queue).  This is the launch that runs BESIDE the sort of the rest of the list: a
// grid of one thread per chunk keeps the command processor placing workgroups for as long as it has chunks left, and the kernels of
// the second stream are not even started until it is through (profiles/r05_overlap_sort_accumulate.txt, A); a grid that is placed in
// one go leaves the dispatcher to them.  (Chunks handed out statically, t, t + threads, ..., ran 15 % slower: ibid., B.)
template <class F, bool PERSIST, int WAVES = (F::N <= 9 ? 4 : 2), bool LDSROW = false, bool SWLDS = false>
__global__ void __launch_bounds__(128, WAVES) k_accumulate(const u32 *__restrict__ bases, const u32 *__restrict__ sorted, const u32 *__restrict__ off,
                                                    u32 *__restrict__ bucket_acc, u32 *__restrict__ parts, u64 stride, unsigned NB, unsigned K,
                                                    unsigned chunks, u32 *__restrict__ long_count, const u32 *__restrict__ stale, AccPart part,
                                                    const u32 *__restrict__ chunk_first)
{
    constexpr int PW = 4 * F::N;
    const unsigned w = blockIdx.y;
    const unsigned t0 = blockIdx.x * blockDim.x + threadIdx.x;
    // the bucket a chunk starts in: handed over by k_chunk_first, or (launches on a part of the list whose offsets are still being
    // written elsewhere: the experimental sort / accumulate overlap) found by binary search
    const u32 *fw = chunk_first ? chunk_first + (u64)w * chunks : nullptr;
    if (t0 == 0) long_count[w] = 0; // the fix-up's queue of long buckets starts empty (it runs behind this kernel on the same stream)
    // the registered buffer no longer holds what was registered (the digits kernel of this range found out): the call will be repeated
    // from the caller's buffer, and nine tenths of the work it would waste are in this kernel
    if (stale && *stale) return;
    const u32 *ow = off + (u64)w * (NB + 1);
    const u32 lo_pos = part.pos ? part.pos[part.lo] : 0u;
    const u32 limit = part.pos ? part.pos[part.hi] : ow[NB]; // pos[cells] is the number of entries, too
    const unsigned search_hi = part.pos ? part.hi_bucket : NB;
    const bool takes_short_chunk = part.pos ? part.last != 0 : true;
    const u32 *sw = sorted + (u64)w * stride;
    u32 *bw = bucket_acc + (u64)w * NB * PW;
    __shared__ uint4 s_rows[LDSROW ? 2 * (2 * F::L / 4) * 64 : 1]; // a wave's region: 2 L / 4 pieces of 1 KB
    uint4 *lds_wave = s_rows + (threadIdx.x >> 6) * (LDSROW ? (2 * F::L / 4) * 64 : 0);
    __shared__ uint4 s_words[SWLDS ? 2 * 512 : 1]; // a wave's region: two sector buffers of 4 pieces x 64 lanes x 16 bytes
    uint4 *lds_words = s_words + (threadIdx.x >> 6) * (SWLDS ? 512 : 0);
    const unsigned lane = threadIdx.x & 63u;
    if constexpr (PERSIST) {
        const unsigned first = lo_pos / K; // the chunks before it end at or below lo_pos: an earlier launch's
#pragma unroll 1
        for (;;) { // every wave leaves when the counter has passed the part's last chunk
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(part.queue, 64u);
            base = __builtin_amdgcn_readfirstlane(base) + first;
            if (base >= chunks || (u64)base * K >= limit) return;
            const unsigned t = base + lane;
            const u32 start = t * K;
            u32 end = start + K;
            bool mine = t < chunks && start < limit;
            bool clamped = false;
            if (end > limit) {
                mine = mine && takes_short_chunk;
                end = limit;
                clamped = true; // the list's last, short chunk: no earlier launch took it (they only take whole chunks), even if it ends AT lo_pos
            }
            if (mine && (end > lo_pos || clamped))
                accumulate_chunk<F, LDSROW, SWLDS>(bases, sw, ow, bw, parts + ((u64)w * chunks + t) * 2 * PW, fw ? fw[t] : owner_bucket(ow, search_hi, start), start, end, lds_wave, lane, lds_words);
        }
    } else {
        const unsigned t = t0;
        if (t >= chunks) return;
        const u32 start = t * K;
        if (start >= limit) return;
        u32 end = start + K;
        bool clamped = false;
        if (end > limit) {
            if (!takes_short_chunk) return; // ends among entries a later launch owns -- or is the list's last, short chunk, which the last launch takes
            end = limit;
            clamped = true;
        }
        // (a short last chunk may end exactly at lo_pos -- everything lies in the earlier launch's part -- and still belongs here: the earlier
        // launch only took whole chunks)
        if (end > lo_pos || clamped)
            accumulate_chunk<F, LDSROW, SWLDS>(bases, sw, ow, bw, parts + ((u64)w * chunks + t) * 2 * PW, fw ? fw[t] : owner_bucket(ow, search_hi, start), start, end, lds_wave, lane, lds_words);
    }
}

// SHARED ROWS: the gathers of a wave, four lanes to a row.  A lane that loads its own 64-byte row issues four 16-byte loads that each
// touch 64 different rows -- 64 different pages of a 12 GiB table --, and what then binds the gather rate is the number of address
// translations per second, not HBM: 20 G rows/s beyond a footprint of ~4 GiB against 56 G rows/s below it, and 13 % of this kernel
// (profiles/r05_accumulate_table_footprint.txt).  Here load instruction m of 2 L / 4 reads pieces 64 m .. 64 m + 63 of the wave's 64 rows
// laid end to end -- piece f belongs to the row of lane f / P --, so one instruction touches 16 rows, and global_load_lds puts piece f at
// 16 f bytes of the wave's LDS region: lane l finds its row at 64 l (tools/ubench_gather_rate.hip: 47.8 G rows/s at 12 GiB this way).
// Every lane of the wave must take part in every fetch, so the loop runs a wave-uniform K iterations and a lane whose chunk is
// shorter (the list's last wave) or absent offers row 0 and does nothing with it.
template <class F>
__device__ __forceinline__ void fetch_rows_shared(uint4 *lds_wave, const u32 *bases, u32 entry, unsigned lane)
{
    constexpr unsigned P = 2 * F::L / 4;
#pragma unroll
    for (unsigned m = 0; m < P; m++) {
        const unsigned f = m * 64 + lane;
        const u32 e = (u32)__shfl((int)entry, (int)(f / P));
        const u32 *src = bases + (u64)(e & PANDA_ROW_MASK) * 2 * F::L + 4 * (f % P);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(lds_wave + m * 64), 16, 0, 0);
    }
}
template <class F>
__device__ __forceinline__ void read_row_shared(PackedBase<F> &b, const uint4 *lds_wave, unsigned lane)
{
    constexpr unsigned P = 2 * F::L / 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // see read_base_lds
#pragma unroll
    for (unsigned j = 0; j < P; j++) {
        const uint4 v = lds_wave[lane * P + j];
        b.w[4 * j] = v.x;
        b.w[4 * j + 1] = v.y;
        b.w[4 * j + 2] = v.z;
        b.w[4 * j + 3] = v.w;
    }
}

template <class F, int WAVES = (F::N <= 9 ? 4 : 2)>
__global__ void __launch_bounds__(128, WAVES) k_accumulate_shared(const u32 *__restrict__ bases, const u32 *__restrict__ sorted, const u32 *__restrict__ off,
                                                           u32 *__restrict__ bucket_acc, u32 *__restrict__ parts, u64 stride, unsigned NB, unsigned K,
                                                           unsigned chunks, u32 *__restrict__ long_count, const u32 *__restrict__ stale)
{
    constexpr int PW = 4 * F::N;
    constexpr unsigned P = 2 * F::L / 4;
    const unsigned w = blockIdx.y;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) long_count[w] = 0;
    if (stale && *stale) return;
    const u32 *ow = off + (u64)w * (NB + 1);
    const u32 *sw = sorted + (u64)w * stride;
    u32 *bw = bucket_acc + (u64)w * NB * PW;
    u32 *pw = parts + ((u64)w * chunks + t) * 2 * PW;
    const u32 limit = ow[NB];
    const bool has = t < chunks && (u64)t * K < limit;
    if (__ballot(has) == 0) return; // wave-uniform
    const u32 start = has ? t * K : 0u;
    const u32 end = has ? (u32)min((u64)start + K, (u64)limit) : 0u;
    __shared__ uint4 s_rows[2 * P * 64];
    uint4 *lds_wave = s_rows + (threadIdx.x >> 6) * (P * 64);
    const unsigned lane = threadIdx.x & 63u;

    u32 b = 0, next = 0;
    bool run_starts_inside = false;
    if (has) {
        b = owner_bucket(ow, NB, start);
        next = ow[b + 1];
        run_starts_inside = ow[b] >= start;
    }
    Xyzz<F> acc;
    xyzz_set_identity(acc);
    const bool quads = has && ((reinterpret_cast<uintptr_t>(sw + start) & 15) == 0);
    uint4 quad = make_uint4(0, 0, 0, 0);
    if (quads) quad = *reinterpret_cast<const uint4 *>(sw + start);
    u32 cur_entry = has ? (quads ? quad.x : sw[start]) : 0u;
    u32 ahead_entry = quads ? quad.y : (start + 1 < end ? sw[start + 1] : 0u);
    fetch_rows_shared<F>(lds_wave, bases, cur_entry, lane);
#pragma unroll 1
    for (u32 i = 0; i < K; i++) { // K is the same for every lane: the fetches below are the whole wave's
        const u32 pos = start + i;
        const bool live = pos < end;
        const u32 entry = cur_entry;
        PackedBase<F> row;
        read_row_shared<F>(row, lds_wave, lane);
        Fe<F> cx, cy;
        bool cinf;
        unpack_base<F, ACC_RAW_Y && RawOperandOk<F>::value>(cx, cy, cinf, row, entry);
        const bool boundary = live && pos >= next;
        if (boundary) { // as in accumulate_chunk: the boundary's dependent loads come before the fetch is issued
            store_xyzz<F>(run_starts_inside ? bw + (u64)b * PW : pw, acc);
            run_starts_inside = true;
            do {
                b++;
                next = ow[b + 1];
            } while (next <= pos);
            if (cinf)
                xyzz_set_identity(acc);
            else {
                Fe<F> ty;
                fe_norm(ty, cy);
                xyzz_from_affine(acc, cx, ty);
            }
        }
        cur_entry = (pos + 1 < end) ? ahead_entry : 0u;
        if (i + 1 < K) fetch_rows_shared<F>(lds_wave, bases, cur_entry, lane);
        if (pos + 2 < end) {
            if (quads) {
                const u32 idx = i + 2;
                if ((idx & 3u) == 0) quad = *reinterpret_cast<const uint4 *>(sw + pos + 2);
                ahead_entry = (idx & 2u) ? ((idx & 1u) ? quad.w : quad.z) : ((idx & 1u) ? quad.y : quad.x);
            } else
                ahead_entry = sw[pos + 2];
        }
        if (!live || boundary || cinf) continue;
        if (xyzz_is_identity(acc)) {
            Fe<F> ty;
            fe_norm(ty, cy);
            xyzz_from_affine(acc, cx, ty);
            continue;
        }
        const int rare = xyzz_madd_core(acc, cx, cy);
        if (rare) {
            if (rare == 1) {
                PackedBase<F> again;
                fetch_base<F>(again, bases, entry);
                unpack_base<F>(cx, cy, cinf, again, entry);
                xyzz_dbl_affine(acc, cx, cy);
            } else
                xyzz_set_identity(acc);
        }
    }
    if (!has) return;
    const bool ends_inside = next <= end;
    u32 *dst = (run_starts_inside && ends_inside) ? bw + (u64)b * PW : (run_starts_inside ? pw + PW : pw);
    store_xyzz<F>(dst, acc);
}

// bucket pieces: a bucket that spans chunks t0 < t1 is the LAST run of t0 (stored in slot 1, or slot 0 if it
// also is t0's first run and started earlier -- impossible here since t0 = start / K), the ONLY run of every
// chunk strictly between (slot 0) and the FIRST run of t1 (slot 0).
// Buckets cut into more than LONG_SPAN pieces (heavily skewed scalars) are queued for k_fixup_long instead of being
// summed by one thread.
constexpr unsigned LONG_SPAN = 128;
constexpr unsigned LONG_BLOCKS = 256; // workgroups per window that serve the queue

// MERGE (point-range chunks, every range after the first): `bucket_acc` holds this range's buckets -- only the non-empty ones are
// defined -- and every non-empty bucket is added into `total`, the running sum over the ranges, here instead of in a pass of its own.
template <class F, bool MERGE>
__global__ void __launch_bounds__(256) k_fixup(const u32 *__restrict__ off, const u32 *__restrict__ parts, u32 *__restrict__ bucket_acc, u32 *__restrict__ total,
                                               unsigned NB, unsigned K, unsigned chunks, u32 *__restrict__ long_count, u32 *__restrict__ long_list, unsigned long_cap)
{
    constexpr int PW = 4 * F::N;
    const unsigned w = blockIdx.y;
    const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= NB) return;
    const u32 *ow = off + (u64)w * (NB + 1);
    const u32 s = ow[b], e = ow[b + 1];
    Xyzz<F> acc, q;
    if (s == e) { // empty: nothing to add to the total; the first range's bucket array is defined here (it is never zero-filled)
        if (!MERGE) {
            xyzz_set_identity(acc);
            store_xyzz<F>(bucket_acc + ((u64)w * NB + b) * PW, acc);
        }
        return;
    }
    const u32 t0 = s / K, t1 = (e - 1) / K;
    if (t0 == t1) { // lies inside one chunk: written by k_accumulate
        if (!MERGE) return;
        load_xyzz<F>(acc, bucket_acc + ((u64)w * NB + b) * PW);
    } else if (t1 - t0 > LONG_SPAN) {
        u32 slot = atomicAdd(&long_count[w], 1u); // a handful per window at most: (t1 - t0) > LONG_SPAN bounds it by chunks / LONG_SPAN
        if (slot < long_cap) {
            u32 *e3 = long_list + ((u64)w * long_cap + slot) * 3;
            e3[0] = b;
            e3[1] = t0;
            e3[2] = t1;
        }
        return;
    } else {
        // (loading the piece of chunk t + 1 under the addition of chunk t's changes nothing: these waves are bound by the additions)
        const u32 *pw = parts + (u64)w * chunks * 2 * PW;
        load_xyzz<F>(acc, pw + ((u64)t0 * 2 + 1) * PW);
        for (u32 t = t0 + 1; t <= t1; t++) {
            load_xyzz<F>(q, pw + (u64)t * 2 * PW);
            xyzz_add_k(acc, q);
        }
    }
    if (MERGE) {
        load_xyzz<F>(q, total + ((u64)w * NB + b) * PW);
        xyzz_add_k(acc, q);
        store_xyzz<F>(total + ((u64)w * NB + b) * PW, acc);
    } else
        store_xyzz<F>(bucket_acc + ((u64)w * NB + b) * PW, acc);
}

// one workgroup per queued bucket: 256 threads stride over its pieces, then an LDS tree; the result replaces dst[b], or is added to it (ADD)
template <class F>
__global__ void __launch_bounds__(256) k_fixup_long(const u32 *__restrict__ parts, u32 *__restrict__ dst, unsigned add, unsigned NB, unsigned chunks,
                                                    const u32 *__restrict__ long_count, const u32 *__restrict__ long_list, unsigned long_cap)
{
    constexpr int PW = 4 * F::N;
    __shared__ __attribute__((aligned(16))) u32 lds[256 * PW];
    const unsigned w = blockIdx.y, t = threadIdx.x;
    const u32 count = min(long_count[w], long_cap);
    const u32 *pw = parts + (u64)w * chunks * 2 * PW;
    for (u32 item = blockIdx.x; item < count; item += gridDim.x) { // count is uniform over the block: barriers below are safe
        const u32 *e3 = long_list + ((u64)w * long_cap + item) * 3;
        const u32 b = e3[0], t0 = e3[1], t1 = e3[2];
        Xyzz<F> acc, q;
        xyzz_set_identity(acc);
        for (u32 c = t0 + t; c <= t1; c += 256) {
            load_xyzz<F>(q, pw + ((u64)c * 2 + (c == t0 ? 1 : 0)) * PW);
            xyzz_add_k(acc, q);
        }
        store_xyzz<F>(lds + t * PW, acc);
        __syncthreads();
        for (unsigned s = 128; s > 0; s >>= 1) {
            if (t < s) {
                load_xyzz<F>(q, lds + (t + s) * PW);
                xyzz_add_k(acc, q);
                store_xyzz<F>(lds + t * PW, acc);
            }
            __syncthreads();
        }
        if (t == 0) {
            if (add) {
                load_xyzz<F>(q, dst + ((u64)w * NB + b) * PW);
                xyzz_add_k(acc, q);
            }
            store_xyzz<F>(dst + ((u64)w * NB + b) * PW, acc);
        }
        __syncthreads();
    }
}

// ---- bucket reduction: sum over b of (b+1) * B_b per list ------------------------------------------------------------
// Replaces the per-bucket c-step double-and-add of msm_cuda.cuh:411-420 (~240 mulmods per bucket).  The bucket index is split
// b = hi * cols + lo (rows = 2^a values of hi, cols = 2^b of lo), so that
//     sum_b (b + 1) B_b  =  sum_hi R_hi  +  cols * sum_hi hi * R_hi  +  sum_lo lo * C_lo,     R_hi = sum_lo B_{hi,lo},  C_lo = sum_hi B_{hi,lo}:
//   k_sum_lines        row sums and column sums are PLAIN sums, two additions per bucket in all and no dependent chain longer than a
//                      line; one wave per line (or per segment of a column, when columns are longer than rows).
//   k_weighted_slots   the two short weighted sums are taken bit by bit: slot 1+j adds up the R_hi (C_lo) whose index has bit j set
//                      and is doubled j + b (j) times, slot 0 adds up all R_hi;  k_slot_total adds the list's slots up and, with
//                      tables, writes the result in wire form.  No scalar multiplications anywhere.
// (Rounds 1-3 first took running sums over groups of 4-8 buckets -- S_g, T_g, two additions per bucket as well -- and split the GROUP
// index: one more kernel and 8-16 more dependent additions in front of the same trees.)
// Everything here is a tree of dependent additions on a few waves: its time is (levels) x (latency of one addition), whatever the
// input size.  A wave's sums therefore run as: every lane adds up its share, the 64 partial sums go through LDS to sixteen quads, and
// the quads finish with four-lane additions (wave_reduce; curve29_quad.h: 3.4 us per level instead of 7.3).
// 64-thread workgroup: lane-wise partial sums `acc` -> their total, replicated over quad 0 (lanes 0..3).  lds: 64 points.
template <class F>
__device__ __forceinline__ void wave_reduce(Xyzz<F> &acc, u32 *lds, unsigned lane)
{
    constexpr int PW = 4 * F::N;
    const unsigned quad = lane >> 2, role = lane & 3u;
    Xyzz<F> q;
    store_xyzz<F>(lds + lane * PW, acc);
    __syncthreads();
    load_xyzz<F>(acc, lds + (4 * quad) * PW);
#pragma unroll 1
    for (unsigned k = 1; k < 4; k++) {
        load_xyzz<F>(q, lds + (4 * quad + k) * PW);
        xyzz_add_q(acc, q, role);
    }
#pragma unroll 1
    for (unsigned s = 8; s > 0; s >>= 1) {
        __syncthreads(); // the slots written below were read above / in the previous level
        if (quad >= s && quad < 2 * s && role == 0) store_xyzz<F>(lds + quad * PW, acc);
        __syncthreads();
        if (quad < s) {
            load_xyzz<F>(q, lds + (quad + s) * PW);
            xyzz_add_q(acc, q, role);
        }
    }
}

// workgroup (x, list), one wave.  x < rows: row x;  else segment (x - rows) / cols of column (x - rows) % cols (csplit segments of
// rows / csplit buckets each: with rows = 2 cols a column is cut in two, so that every line is cols buckets long)
template <class F>
__global__ void __launch_bounds__(64) k_sum_lines(const u32 *__restrict__ bucket_acc, u32 *__restrict__ outR, u32 *__restrict__ outC, unsigned rows, unsigned cols,
                                                  unsigned csplit)
{
    constexpr int PW = 4 * F::N;
    __shared__ __attribute__((aligned(16))) u32 lds[64 * PW];
    const unsigned x = blockIdx.x, list = blockIdx.y, lane = threadIdx.x;
    const u64 NB = (u64)rows * cols;
    const u32 *src;
    u32 *dst;
    unsigned count, stride;
    if (x < rows) {
        src = bucket_acc + (list * NB + (u64)x * cols) * PW;
        count = cols;
        stride = 1;
        dst = outR + ((u64)list * rows + x) * PW;
    } else {
        const unsigned seg = (x - rows) / cols, lo = (x - rows) % cols;
        count = rows / csplit;
        src = bucket_acc + (list * NB + (u64)seg * count * cols + lo) * PW;
        stride = cols;
        dst = outC + ((u64)list * cols * csplit + (x - rows)) * PW;
    }
    // (measured and dropped: the next bucket loaded under the addition of the current one, and every line starting its sweep at a
    // different place -- whichever of the row and column passes runs first after the fix-up takes 1.4x as long as the other)
    Xyzz<F> acc, q;
    xyzz_set_identity(acc);
#pragma unroll 1
    for (unsigned i = lane; i < count; i += 64) {
        load_xyzz<F>(q, src + (u64)i * stride * PW);
        xyzz_add_k(acc, q);
    }
    wave_reduce<F>(acc, lds, lane);
    if (lane == 0) store_xyzz<F>(dst, acc);
}

// workgroup (slot, list, part), one wave.  slot 0: sum of the R_hi;  slot 1+j (j < a): the R_hi with bit j of hi set, doubled j + b
// times;  slot 1+a+j (j < b): the column (segment) sums with bit j of lo set, doubled j times (rows = 2^a, cols = 2^b).  A slot's
// entries are shared out over `parts` workgroups, each of which doubles its own partial sum (2^k (x + y) = 2^k x + 2^k y: the doublings
// of the parts run side by side).  k_slot_total adds the list's slots * parts partial sums up.
template <class F>
__global__ void __launch_bounds__(64) k_weighted_slots(const u32 *__restrict__ inR, const u32 *__restrict__ inC, u32 *__restrict__ slot_out, unsigned a, unsigned b,
                                                       unsigned csplit)
{
    constexpr int PW = 4 * F::N;
    __shared__ __attribute__((aligned(16))) u32 lds[64 * PW];
    const unsigned slot = blockIdx.x, list = blockIdx.y, part = blockIdx.z, parts = gridDim.z, lane = threadIdx.x, role = lane & 3u;
    const unsigned rows = 1u << a, cols = 1u << b, slots = 1 + a + b, partials = slots * parts;
    Xyzz<F> acc, q;
    xyzz_set_identity(acc);
    unsigned doublings = 0;
    if (slot == 0) {
        const u32 *src = inR + (u64)list * rows * PW;
        const unsigned per = (rows + parts - 1) / parts, end = min(rows, (part + 1) * per);
#pragma unroll 1
        for (unsigned i = part * per + lane; i < end; i += 64) {
            load_xyzz<F>(q, src + (u64)i * PW);
            xyzz_add_k(acc, q);
        }
    } else {
        const bool row_slot = slot <= a;
        const unsigned j = row_slot ? slot - 1 : slot - 1 - a;
        // entries: rows row sums, or csplit * cols column-segment sums (entry e belongs to column e % cols: bit j of e is bit j of lo)
        const unsigned half = (row_slot ? rows : cols * csplit) >> 1; // entries whose index has bit j set
        const u32 *src = row_slot ? inR + (u64)list * rows * PW : inC + (u64)list * cols * csplit * PW;
        const u32 low = (1u << j) - 1;
        const unsigned per = (half + parts - 1) / parts, end = min(half, (part + 1) * per);
        doublings = j + (row_slot ? b : 0u);
#pragma unroll 1
        for (unsigned i = part * per + lane; i < end; i += 64) {
            const unsigned idx = ((i & ~low) << 1) | (1u << j) | (i & low);
            load_xyzz<F>(q, src + (u64)idx * PW);
            xyzz_add_k(acc, q);
        }
    }
    wave_reduce<F>(acc, lds, lane);
    if (lane < 4) {
#pragma unroll 1
        for (unsigned d = 0; d < doublings; d++) {
            xyzz_dbl_q(q, acc, role);
            acc = q;
        }
        if (lane == 0) store_xyzz<F>(slot_out + ((u64)list * partials + slot * parts + part) * PW, acc);
    }
}

// workgroup `list`, one wave: the list's slots * parts partial sums -> win[list] -- as an XYZZ point (emit = 0: the host still has a
// Horner step to do over the lists), or, for a single list whose sum IS the result, in the wire format of the C ABI: Jacobian
// X || Y || Z (emit = 1) or homogeneous (emit = 2).  `win` may be device memory or pinned host memory.
// (A kernel of its own rather than the last-to-finish workgroup of k_weighted_slots: the device-scope fences such a hand-off needs
// write the XCD's L2 back in every workgroup -- profiles/r04_msm_small_sizes.txt -- and cost more than the launch.)
template <class F>
__global__ void __launch_bounds__(64) k_slot_total(const u32 *__restrict__ slot_out, u32 *__restrict__ win, unsigned partials, unsigned emit)
{
    constexpr int PW = 4 * F::N;
    __shared__ __attribute__((aligned(16))) u32 lds[64 * PW];
    const unsigned list = blockIdx.x, lane = threadIdx.x;
    Xyzz<F> acc, q;
    xyzz_set_identity(acc);
#pragma unroll 1
    for (unsigned i = lane; i < partials; i += 64) {
        load_xyzz<F>(q, slot_out + ((u64)list * partials + i) * PW);
        xyzz_add_k(acc, q);
    }
    wave_reduce<F>(acc, lds, lane);
    if (lane == 0) {
        if (emit == 0)
            store_xyzz<F>(win + (u64)list * PW, acc);
        else {
            u32 out[3 * F::L];
            if (emit == 2)
                xyzz_to_homogeneous_wire(out, acc);
            else
                xyzz_to_jacobian_wire(out, acc);
#pragma unroll
            for (int k = 0; k < 3 * F::L; k++) win[k] = out[k];
        }
    }
}

// Precomputed window tables for cached bases (the reference left the idea as a stub,
// msm_host.cuh:248-265): row i of table k+1 is 2^width[k] times row i of table k, back in affine form, so that
// digit d_k of scalar i is served by the single addition  d_k * T_k[i]  into a bucket space shared by all windows.
template <class F>
__global__ void __launch_bounds__(128) k_table_step(const u32 *__restrict__ prev, u32 *__restrict__ next, u64 n, unsigned steps)
{
    constexpr int L = F::L;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PackedBase<F> b;
    load_words<2 * L>(b.w, prev + i * 2 * L);
    Fe<F> x, y;
    bool inf;
    unpack_base<F>(x, y, inf, b, 0u);
    u32 o[2 * L];
    if (inf) {
#pragma unroll
        for (int k = 0; k < 2 * L; k++) o[k] = 0;
    } else {
        Xyzz<F> p, d;
        xyzz_dbl_affine(p, x, y);
#pragma unroll 1
        for (unsigned s = 1; s < steps; s++) { // the group has odd order: a doubling never reaches the identity
            xyzz_dbl_k(d, p);
            p = d;
        }
        xyzz_to_affine_internal(x, y, p);
        fe_reduce_once(x);
        fe_reduce_once(y);
        fe_pack(o, x);
        fe_pack(o + L, y);
    }
    store_words<2 * L>(next + i * 2 * L, o);
}

template <class F>
void host_horner(Xyzz<F> &result, const std::vector<Xyzz<F>> &windows, const panda::WindowPlan &plan)
{
    Xyzz<F> acc, d;
    xyzz_set_identity(acc);
    for (int w = (int)windows.size() - 1; w >= 0; w--) {
        for (unsigned k = 0; k < plan.width[w]; k++) {
            xyzz_dbl(d, acc);
            acc = d;
        }
        xyzz_add(acc, windows[w]);
    }
    result = acc;
}

unsigned floor_log2(u64 v)
{
    unsigned l = 0;
    while (v >> (l + 1)) l++;
    return l;
}

template <class C>
hipError_t msm_execute(const panda_msm_configuration &cfg, const panda::MsmRegistration *registration, panda::MsmTuning tuning, float *phase_ms, bool *stale,
                       const panda::MsmPipeline *pipe)
{
    typedef typename C::Fq Fq;
    constexpr int PW = 4 * Fq::N;
    constexpr int LQ = Fq::L;
    constexpr unsigned curve = panda::msm_scalar_field_of(C::ID); // the sort and the window plans only care about the scalar field
    hipStream_t stream = static_cast<hipStream_t>(cfg.stream.handle);
    const unsigned log_n = cfg.log_scalars_count;
    if (log_n > 26 || !cfg.bases || !cfg.scalars || !cfg.results) return hipErrorInvalidValue;
    const u64 n = (u64)1 << log_n;

    const bool registered = registration != nullptr;
    const bool tabled = registered && registration->tabled;
    const panda::WindowPlan plan = tabled ? registration->plan : panda::make_safe_window_plan(curve, tuning.window_bits);
    const unsigned W = plan.W;
    const unsigned c = plan.width[0]; // widest window
    const unsigned NB = 1u << (c - 1);
    const unsigned lists = tabled ? 1u : W; // independent bucket spaces
    // Point ranges: only against registered bases (the converted rows / tables exist for every range already), ranges of at least
    // 2^16 points, and only where the three-level sort has a geometry for the range sizes.  R ranges hold n/2^(R-1), n/2^(R-1),
    // n/2^(R-2), ..., n/2 points: the first upload -- the only one nothing runs beside -- is short, and every later range is at most
    // twice its predecessor, so its upload (PCIe moves a range about 1.9x faster than the kernels consume one) hides behind it.
    // EQUAL ranges (pipe->ranges = 0x100 | R, R a power of two; resident scalars): R ranges of n / R points, one after the other on the
    // caller's stream.  Not for uploads -- for the table footprint: a range gathers rows of W tables x n / R points, and the memory system
    // serves random 64-byte rows 2.5x faster below ~3.5 GiB of footprint than above (profiles/r05_ubench_gather_rate.txt).
    const bool equal_ranges = pipe && registered && (pipe->ranges & 0x100u) && !pipe->h_scalars;
    unsigned eq_log = 0;
    while (equal_ranges && (2u << eq_log) <= (pipe->ranges & 0xffu)) eq_log++;
    unsigned nranges = equal_ranges ? (1u << eq_log) : ((pipe && registered) ? std::min(std::max(pipe->ranges & 0xffu, 1u), 8u) : 1u);
    auto ranges_ok = [&](unsigned R) {
        if (equal_ranges) return log_n >= eq_log + 16 && (!tabled || panda::msm_sort_tabled_supported(log_n - eq_log, plan));
        if (log_n < (R - 1) + 16) return false;
        for (unsigned lc = log_n - (R - 1); tabled && lc < log_n; lc++)
            if (!panda::msm_sort_tabled_supported(lc, plan)) return false;
        return true;
    };
    if (equal_ranges) {
        while (eq_log > 0 && !ranges_ok(1u << eq_log)) eq_log--;
        nranges = 1u << eq_log;
    } else
        while (nranges > 1 && !ranges_ok(nranges)) nranges--;
    auto range_log = [&](unsigned r) { return nranges == 1 ? log_n : (equal_ranges ? log_n - eq_log : log_n - (r == 0 ? nranges - 1 : nranges - r)); };
    auto range_row0 = [&](unsigned r) {
        if (equal_ranges) return (u64)r << (log_n - eq_log);
        return (nranges == 1 || r == 0) ? (u64)0 : (u64)1 << (log_n - (nranges - r));
    };
    struct RangeGeom {
        u64 stride;               // entries per list (upper bound)
        unsigned K, chunks, long_cap; // sorted entries per accumulate thread, accumulate threads per list, queue slots for long buckets
    };
    auto range_geom = [&](unsigned log_c) {
        RangeGeom g;
        g.stride = tabled ? (u64)W << log_c : (u64)1 << log_c;
        // sorted entries per accumulate thread: as many as leave about 2^20 threads over all lists -- six rounds of the chip's
        // 2^18 resident threads at four waves per SIMD, so neither a half-empty chip nor a half-empty last round costs much
        // (a 2^21-point range at K = 128 kept three waves per SIMD busy: 2.4 ms instead of 1.8); 64 ... 256 measure the same at 2^24
        const u64 per_k = ((u64)lists * g.stride) >> 20;
        g.K = per_k >= 128 ? 128 : (per_k >= 64 ? 64 : (per_k >= 32 ? 32 : 16));
        // below 2^22 points the threads are fewer than four rounds of the chip and the best chunk is a matter of how the last round fills:
        // interleaved sweeps (tools/chunk_sweep.py, profiles/r05_chunk_sweep.txt) put 24 in front at 2^19 (-3.4 %) and 32 at 2^20 / 2^21 (-1 %)
        if (Fq::N <= 9 && per_k < 32) g.K = per_k >= 12 ? 32 : (per_k >= 6 ? 24 : 16);
        if (tuning.chunk) g.K = std::min(std::max((tuning.chunk + 3u) & ~3u, 4u), 1024u); // multiples of four: chunks start on 16 bytes
        g.chunks = (unsigned)((g.stride + g.K - 1) / g.K);
        g.long_cap = g.chunks / LONG_SPAN + 2;
        return g;
    };
    // row / column split of the bucket index: b = hi * cols + lo, rows = 2^a >= cols = 2^b; columns of 2 cols buckets are summed in two halves
    const unsigned rc_b = (c - 1) / 2, rc_a = (c - 1) - rc_b;
    const unsigned rc_rows = 1u << rc_a, rc_cols = 1u << rc_b, rc_csplit = rc_a > rc_b ? 2u : 1u;
    const unsigned slots = 1 + rc_a + rc_b; // slot 0: all rows; then one per bit of hi and of lo
    // workgroups per slot of k_weighted_slots: at most four entries per lane (2^21 buckets: 1024 of the 2048 row sums per slot)
    const unsigned rc_parts = std::max(1u, std::max(rc_rows, rc_cols * rc_csplit) / 512u);

    // ---- scratch
    const size_t sz_bases = registered ? 0 : panda::align256(n * 2 * LQ * 4);
    size_t sz_sort = 0, sz_parts = 0, sz_llist = 0; // maxima over the range sizes in use (neither is monotonic in the size by construction)
    for (unsigned r = 0; r < nranges; r++) {
        const unsigned lc = range_log(r);
        const RangeGeom g = range_geom(lc);
        sz_sort = std::max(sz_sort, tabled ? panda::msm_sort_tabled_bytes(lc, plan) : panda::msm_sort_plain_bytes(lc, plan));
        sz_parts = std::max(sz_parts, panda::align256((size_t)lists * g.chunks * 2 * PW * 4));
        sz_llist = std::max(sz_llist, panda::align256((size_t)lists * g.long_cap * 3 * 4));
    }
    size_t sz_first = 0;
    for (unsigned r = 0; r < nranges; r++) sz_first = std::max(sz_first, panda::align256((size_t)lists * range_geom(range_log(r)).chunks * 4));
    const size_t sz_bacc = panda::align256((size_t)lists * NB * PW * 4);
    const size_t sz_l1 = panda::align256((size_t)lists * (rc_rows + rc_cols * rc_csplit) * PW * 4);
    const size_t sz_win = 256; // the stale-registration flag (device copy: k_accumulate reads it)
    const size_t sz_slots = panda::align256((size_t)lists * slots * rc_parts * PW * 4);
    const size_t sz_lcount = panda::align256((size_t)lists * 4);
    panda::Arena &arena = panda::thread_arena();
    // Ranges alternate between two lanes -- the caller's stream and a helper stream of this host thread -- each with its own sort
    // scratch, pieces and range buckets, so that range r+1 is sorted (LDS / HBM work) while range r is still being accumulated
    // (vector issue); the fix-ups, which all add into the one total, are chained by events.
    const unsigned lanes = (nranges > 1 && !equal_ranges) ? 2u : 1u; // equal ranges run one after the other: side by side they would share the footprint again
    PANDA_TRY(arena.reserve(sz_bases + sz_bacc + lanes * (sz_sort + sz_bacc + sz_parts + sz_lcount + sz_llist + sz_first + 1280) + sz_l1 + sz_win + sz_slots + 8192));
    const u32 *d_bases = registered ? (const u32 *)registration->converted : (const u32 *)arena.take(sz_bases);
    u32 *d_bacc = (u32 *)arena.take(sz_bacc);
    u32 *d_bacc_range[2] = {d_bacc, d_bacc}, *d_parts_l[2] = {nullptr, nullptr}, *d_lcount_l[2] = {nullptr, nullptr}, *d_llist_l[2] = {nullptr, nullptr};
    u32 *d_first_l[2] = {nullptr, nullptr};
    for (unsigned l = 0; l < lanes; l++) {
        if (nranges > 1) d_bacc_range[l] = (u32 *)arena.take(sz_bacc); // buckets of the range in flight on this lane, added into d_bacc by its fix-up
        d_parts_l[l] = (u32 *)arena.take(sz_parts);
        d_lcount_l[l] = (u32 *)arena.take(sz_lcount);
        d_llist_l[l] = (u32 *)arena.take(sz_llist);
        d_first_l[l] = (u32 *)arena.take(sz_first);
        if (!d_bacc_range[l] || !d_parts_l[l] || !d_lcount_l[l] || !d_llist_l[l] || !d_first_l[l]) return hipErrorOutOfMemory;
    }
    u32 *d_l1 = (u32 *)arena.take(sz_l1);
    u32 *d_stale = (u32 *)arena.take(sz_win);
    u32 *d_slots = (u32 *)arena.take(sz_slots);
    if (!d_bases || !d_bacc || !d_l1 || !d_stale || !d_slots) return hipErrorOutOfMemory;
    // What the call hands back to this host thread travels through pinned host memory the kernels write into directly: word 0 is set
    // by the digits kernel when the registered buffer has changed, the window sums (or, with tables, the finished result) land behind it.
    u32 *mail = nullptr;
    PANDA_TRY(panda::thread_mailbox(&mail));
    if (64 + (size_t)lists * PW > panda::MAILBOX_WORDS) return hipErrorInvalidValue;
    volatile u32 *h_flag = mail;
    u32 *h_win = mail + 64;
    *h_flag = 0;
    const panda::SampleCheck sample_check{registered ? (const u32 *)cfg.bases : nullptr, registered ? registration->samples : nullptr, n, 2u * LQ, d_stale, mail};
    // With tables the single list's sum is the result, and the last kernel writes it in wire form where the caller wants it -- if that
    // is memory a kernel can address (device memory of this device, pinned or registered host memory; unit.rs:32-47, msm_test.cu:53,125);
    // anything else gets it through the mailbox and a copy.
    u32 *res_dev = nullptr;
    if (tabled) {
        hipPointerAttribute_t at{};
        int dev = -1;
        if (hipPointerGetAttributes(&at, cfg.results) == hipSuccess && hipGetDevice(&dev) == hipSuccess && at.devicePointer &&
            ((at.type == hipMemoryTypeDevice && at.device == dev) || at.type == hipMemoryTypeHost))
            res_dev = (u32 *)at.devicePointer;
        else
            (void)hipGetLastError();
    }
    // clock stamps around the accumulation of the last range (panda_set_clock_stamps; panda_internal.h)
    const bool stamps = panda::clock_stamps_enabled();
    uint64_t *stamp_block = nullptr;
    if (stamps) {
        PANDA_TRY(panda::thread_stamp_blocks(&stamp_block));
        for (unsigned i = 0; i < 2 * 2 * panda::CLOCK_STAMP_SLOTS; i++) stamp_block[i] = 0;
    }
    panda::thread_msm_clock() = panda::ClockDelta{};
    const size_t sort_mark[2] = {arena.used, arena.used + panda::align256(sz_sort) + 512}; // a lane's sorts carve their scratch from its mark again
    hipStream_t lane_stream[2] = {stream, stream};
    if (lanes > 1) PANDA_TRY(panda::thread_helper_stream(&lane_stream[1]));
    // One call, whole input at once, with tables: levels 2 and 3 of the sort for the rest of the bucket space run on the helper stream
    // beside the accumulation of the front (SortSplit, msm_sort.h).  The two ordering events live as long as the host thread.
    struct SplitEvents {
        hipEvent_t ev[2] = {};
        int device = -1;
        void drop()
        {
            for (auto &e : ev) {
                if (e) (void)hipEventDestroy(e);
                e = nullptr;
            }
        }
        ~SplitEvents() { drop(); }
    };
    static thread_local SplitEvents split_events;
    panda::SortSplit split{};
    const bool want_split = tabled && nranges == 1 && tuning.overlap_front != 0 && tuning.overlap_front < 128;
    // workgroups per CU of the accumulation that runs beside the sort: the kernel's own occupancy (8 x 2 waves of 104 registers for the
    // 9-limb fields with the row staged in LDS, 4 x 2 waves of ~176 for the 14-limb fields): both leave the sort's workgroups room
    const unsigned overlap_wgs = tuning.overlap_wgs ? tuning.overlap_wgs : (Fq::N <= 9 ? 8u : 4u);
    if (want_split) {
        int dev = -1;
        PANDA_TRY(hipGetDevice(&dev));
        if (split_events.device != dev) {
            split_events.drop();
            for (auto &e : split_events.ev) PANDA_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            split_events.device = dev;
        }
        PANDA_TRY(panda::thread_helper_stream(&split.helper));
        split.front_of_128 = tuning.overlap_front;
        split.ev_front = split_events.ev[0];
        split.rest_done = split_events.ev[1];
    }

    struct PhaseEvents { // the per-range events are destroyed on every exit path
        std::vector<hipEvent_t> uploaded, fixed; // range r: its scalars have arrived / its buckets are in the total
        hipEvent_t started = nullptr;
        ~PhaseEvents()
        {
            for (auto &e : uploaded)
                if (e) (void)hipEventDestroy(e);
            for (auto &e : fixed)
                if (e) (void)hipEventDestroy(e);
            if (started) (void)hipEventDestroy(started);
        }
    } phase_events;
    // the eight phase-timer events are kept per host thread and device: creating and destroying them in every call was a
    // measurable share of the host time of a 2^20-point call
    struct TimerEvents {
        hipEvent_t ev[8] = {};
        int device = -1;
        void drop()
        {
            for (auto &e : ev) {
                if (e) (void)hipEventDestroy(e);
                e = nullptr;
            }
        }
        ~TimerEvents() { drop(); }
    };
    static thread_local TimerEvents timer_events;
    {
        int dev = -1;
        PANDA_TRY(hipGetDevice(&dev));
        if (timer_events.device != dev) {
            timer_events.drop();
            for (auto &e : timer_events.ev) PANDA_TRY(hipEventCreate(&e));
            timer_events.device = dev;
        }
    }
    hipEvent_t(&ev)[8] = timer_events.ev;
    // which of the eight events a call records: an event between two kernels keeps the GPU idle for ~6 us (rocprofv3 timeline of a
    // 2^20-point call, profiles/r04_*), so only the level the caller asked for is paid for
    const unsigned timing = tuning.timing;
    auto wanted = [&](int i) { return timing >= 2 || (timing == 1 && (i == 0 || i == 3 || i == 4 || i >= 6)); };
    auto mark = [&](int i) { return wanted(i) ? hipEventRecord(ev[i], stream) : hipSuccess; };

    // upload of range r on the copy stream (a pageable source makes the call block until the range is staged, a pinned one returns at once)
    const char *h_scalars = pipe ? (const char *)pipe->h_scalars : nullptr;
    if (h_scalars) {
        phase_events.uploaded.assign(nranges, nullptr);
        for (auto &e : phase_events.uploaded) PANDA_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    auto upload = [&](unsigned r) -> hipError_t {
        const size_t first = (size_t)range_row0(r) * 32, bytes = ((size_t)1 << range_log(r)) * 32;
        PANDA_TRY(hipMemcpyAsync((char *)cfg.scalars + first, h_scalars + first, bytes, hipMemcpyHostToDevice, pipe->h2d));
        return hipEventRecord(phase_events.uploaded[r], pipe->h2d);
    };

    PANDA_TRY(mark(0));
    if (h_scalars) PANDA_TRY(upload(0));
    if (!registered)
        hipLaunchKernelGGL(k_convert_bases<Fq>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const u32 *)cfg.bases, const_cast<u32 *>(d_bases), n);
    if (lanes > 1) { // the helper lane starts after everything the caller's stream held before this call
        phase_events.fixed.assign(nranges, nullptr);
        for (auto &e : phase_events.fixed) PANDA_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        PANDA_TRY(hipEventCreateWithFlags(&phase_events.started, hipEventDisableTiming));
        PANDA_TRY(hipEventRecord(phase_events.started, stream));
        PANDA_TRY(hipStreamWaitEvent(lane_stream[1], phase_events.started, 0));
    }
    for (unsigned r = 0; r < nranges; r++) { // phases 1..4 of a call in ranges are timed on its last (largest) range
        const bool last = r + 1 == nranges;
        const unsigned lane = lanes > 1 ? (r & 1u) : 0u;
        hipStream_t ls = lane_stream[lane];
        const unsigned log_c = range_log(r);
        const u64 row0 = range_row0(r);
        const RangeGeom g = range_geom(log_c);
        if (h_scalars) PANDA_TRY(hipStreamWaitEvent(ls, phase_events.uploaded[r], 0));
        arena.used = sort_mark[lane];
        panda::SortResult sorted{};
        const panda::SortEvents sort_events{last && wanted(1) ? ev[1] : nullptr, last && wanted(2) ? ev[2] : nullptr};
        const panda::SortPlacement place{nranges > 1 ? log_n : 0u, (uint32_t)row0};
        const void *scalars_r = (const char *)cfg.scalars + row0 * 32;
        if (tabled)
            PANDA_TRY(panda::msm_sort_tabled(ls, arena, curve, scalars_r, log_c, plan, sort_events, &sorted, place, sample_check, want_split ? &split : nullptr));
        else
            PANDA_TRY(panda::msm_sort_plain(ls, arena, curve, scalars_r, log_c, plan, sort_events, &sorted, place, sample_check));
        if (sorted.lists != lists || sorted.NB != NB || sorted.stride != g.stride) return hipErrorInvalidValue;
        if (h_scalars && !last) PANDA_TRY(upload(r + 1)); // behind this range's sort in host order, beside its kernels on the device
        if (last && wanted(3)) PANDA_TRY(hipEventRecord(ev[3], ls));
        if (last && stamps) PANDA_TRY(panda::enqueue_clock_stamp(ls, stamp_block));
        // the first range accumulates straight into the total (zeroed: empty buckets must read as the identity); a later range into
        // its lane's own array, of which only the non-empty buckets are ever read, by the fix-up that adds them to the total
        u32 *target = r == 0 ? d_bacc : d_bacc_range[lane];
        u32 *d_parts = d_parts_l[lane], *d_lcount = d_lcount_l[lane], *d_llist = d_llist_l[lane];
        // every chunk's first bucket, for the launches that see the whole list (not for the split launches of the overlap experiment,
        // whose later offsets are still being written when the front is accumulated)
        const u32 *d_first = nullptr;
        if (tuning.chunk_first && !split.active && tuning.acc_variant != 3) {
            hipLaunchKernelGGL(k_chunk_first, dim3((NB + 255) / 256, lists), dim3(256), 0, ls, sorted.off, d_first_l[lane], NB, g.K, g.chunks);
            d_first = d_first_l[lane];
        }
        // (no zero-fill of the bucket array: the first range's fix-up writes the identity into its empty buckets; k_accumulate empties the
        // fix-up's queue of long buckets)
        if (split.active) {
            // the front of the list while the helper stream sorts the rest -- as the ordinary grid (overlap_wgs = 64) or as a few workgroups per
            // CU that draw their chunks from a counter and leave the second stream room --, then the rest
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, registration->device);
            const unsigned all_wgs = (g.chunks + 127) / 128, persist_wgs = std::min((unsigned)cus * overlap_wgs, all_wgs);
            u32 *d_queue = d_stale + 16; // a word of the flag block
            const AccPart front{split.pos, 0u, split.cut_cell, 0u, split.cut_bucket, d_queue};
            if (persist_wgs == all_wgs) {
                // the ordinary grid; for the 9-limb fields the kernel with its row staged in LDS: 104 registers x 4 waves per SIMD, as fast as the
                // built-in one (profiles/r05_accumulate_lds_row.txt), leave 96 registers per SIMD to the second stream's four-wave workgroups
                if constexpr (Fq::N <= 9 && !IsExt2<Fq>::value)
                    hipLaunchKernelGGL((k_accumulate<Fq, false, 4, true>), dim3(all_wgs, 1), dim3(128), 0, ls, d_bases, sorted.sorted, sorted.off, target, d_parts, g.stride, NB, g.K,
                                       g.chunks, d_lcount, registered ? d_stale : nullptr, front, (const u32 *)nullptr);
                else
                    hipLaunchKernelGGL((k_accumulate<Fq, false>), dim3(all_wgs, 1), dim3(128), 0, ls, d_bases, sorted.sorted, sorted.off, target, d_parts, g.stride, NB, g.K,
                                       g.chunks, d_lcount, registered ? d_stale : nullptr, front, (const u32 *)nullptr);
            } else {
                PANDA_TRY(hipMemsetAsync(d_queue, 0, 4, ls));
                if constexpr (Fq::N <= 9 && !IsExt2<Fq>::value)
                    // the kernel with its row staged in LDS, built for 96 registers: 8 workgroups per CU (4 waves per SIMD) leave 128 registers per
                    // SIMD and 96 KB of LDS to the sort's workgroups (of which only the four-wave ones are placed beside it: profiles/r05_overlap_*)
                    hipLaunchKernelGGL((k_accumulate<Fq, true, 5, true>), dim3(persist_wgs, 1), dim3(128), 0, ls, d_bases, sorted.sorted, sorted.off, target, d_parts, g.stride, NB,
                                       g.K, g.chunks, d_lcount, registered ? d_stale : nullptr, front, (const u32 *)nullptr);
                else
                    hipLaunchKernelGGL((k_accumulate<Fq, true>), dim3(persist_wgs, 1), dim3(128), 0, ls, d_bases, sorted.sorted, sorted.off, target, d_parts, g.stride, NB, g.K,
                                       g.chunks, d_lcount, registered ? d_stale : nullptr, front, (const u32 *)nullptr);
            }
            // the rest is accumulated on the HELPER stream, straight behind its sort: the two accumulate launches touch disjoint chunks, pieces and
            // buckets, so the second one's workgroups fill the CUs as the first one's last round drains (on one stream every launch boundary
            // costs about half a workgroup's lifetime -- 128 additions, 2.3 ms -- of a half-empty chip); the fix-up waits for both
            hipLaunchKernelGGL((k_accumulate<Fq, false>), dim3((g.chunks + 127) / 128, 1), dim3(128), 0, split.helper, d_bases, sorted.sorted, sorted.off, target, d_parts, g.stride, NB,
                               g.K, g.chunks, d_lcount, registered ? d_stale : nullptr, AccPart{split.pos, split.cut_cell, split.cells, 1u, NB, nullptr}, (const u32 *)nullptr);
            PANDA_TRY(hipEventRecord(split.rest_done, split.helper));
            PANDA_TRY(hipStreamWaitEvent(ls, split.rest_done, 0));
        } else if (Fq::N <= 9 && !IsExt2<Fq>::value && tuning.acc_variant == 1) {
            if constexpr (Fq::N <= 9 && !IsExt2<Fq>::value) // five waves per SIMD, the next row staged in LDS (experiment: panda_msm_set_accumulate_variant)
                hipLaunchKernelGGL((k_accumulate<Fq, false, 5, true>), dim3((g.chunks + 127) / 128, lists), dim3(128), 0, ls, d_bases, sorted.sorted, sorted.off, target, d_parts,
                                   g.stride, NB, g.K, g.chunks, d_lcount, registered ? d_stale : nullptr, AccPart{nullptr, 0u, 0u, 1u, NB, nullptr}, d_first);
        } else if (Fq::N <= 9 && !IsExt2<Fq>::value && tuning.acc_variant == 2) {
            if constexpr (Fq::N <= 9 && !IsExt2<Fq>::value) // four waves per SIMD with the LDS-staged row
                hipLaunchKernelGGL((k_accumulate<Fq, false, 4, true>), dim3((g.chunks + 127) / 128, lists), dim3(128), 0, ls, d_bases, sorted.sorted, sorted.off, target, d_parts,
                                   g.stride, NB, g.K, g.chunks, d_lcount, registered ? d_stale : nullptr, AccPart{nullptr, 0u, 0u, 1u, NB, nullptr}, d_first);
        } else if (!IsExt2<Fq>::value && (tuning.acc_variant == 4 || (tuning.acc_variant == 0 && Fq::N <= 9 && g.K % 16 == 0))) {
            // the sorted words in whole 64-byte sectors through LDS: the built-in kernel of the 9-limb fields from round 6 on (k_accumulate's
            // fetch 16.2 -> 13.7 GB per launch at 2^24, +0.5 % clock, -0.5 % time; neutral for the 14-limb fields: variant 4 forces it, 5 forbids it)
            hipLaunchKernelGGL((k_accumulate<Fq, false, (Fq::N <= 9 ? 4 : 2), false, true>), dim3((g.chunks + 127) / 128, lists), dim3(128), 0, ls, d_bases, sorted.sorted, sorted.off,
                               target, d_parts, g.stride, NB, g.K, g.chunks, d_lcount, registered ? d_stale : nullptr, AccPart{nullptr, 0u, 0u, 1u, NB, nullptr}, d_first);
        } else if (tuning.acc_variant == 3) { // the wave's gathers four lanes to a row
            hipLaunchKernelGGL((k_accumulate_shared<Fq>), dim3((g.chunks + 127) / 128, lists), dim3(128), 0, ls, d_bases, sorted.sorted, sorted.off, target, d_parts, g.stride, NB,
                               g.K, g.chunks, d_lcount, registered ? d_stale : nullptr);
        } else
            hipLaunchKernelGGL((k_accumulate<Fq, false>), dim3((g.chunks + 127) / 128, lists), dim3(128), 0, ls, d_bases, sorted.sorted, sorted.off, target, d_parts, g.stride,
                               NB, g.K, g.chunks, d_lcount, registered ? d_stale : nullptr, AccPart{nullptr, 0u, 0u, 1u, NB, nullptr}, d_first);
        if (last && stamps) PANDA_TRY(panda::enqueue_clock_stamp(ls, stamp_block + 2 * panda::CLOCK_STAMP_SLOTS));
        if (last && wanted(4)) PANDA_TRY(hipEventRecord(ev[4], ls));
        // 256-thread workgroups: at 2^16 buckets that is one per CU, a wave per SIMD (with 128 the dispatcher doubled them up on half
        // the CUs and every addition took 1.6x as long: fix-up 0.225 -> 0.162 ms at 2^20 points)
        constexpr unsigned fx_block = 256;
        if (r == 0)
            hipLaunchKernelGGL((k_fixup<Fq, false>), dim3((NB + fx_block - 1) / fx_block, lists), dim3(fx_block), 0, ls, sorted.off, d_parts, target, d_bacc, NB, g.K, g.chunks, d_lcount,
                               d_llist, g.long_cap);
        else {
            if (lanes > 1) PANDA_TRY(hipStreamWaitEvent(ls, phase_events.fixed[r - 1], 0)); // the total is complete up to the previous range
            hipLaunchKernelGGL((k_fixup<Fq, true>), dim3((NB + fx_block - 1) / fx_block, lists), dim3(fx_block), 0, ls, sorted.off, d_parts, target, d_bacc, NB, g.K, g.chunks, d_lcount,
                               d_llist, g.long_cap);
        }
        hipLaunchKernelGGL(k_fixup_long<Fq>, dim3(LONG_BLOCKS, lists), dim3(256), 0, ls, d_parts, d_bacc, r == 0 ? 0u : 1u, NB, g.chunks, d_lcount, d_llist, g.long_cap);
        if (lanes > 1) PANDA_TRY(hipEventRecord(phase_events.fixed[r], ls));
        PANDA_TRY(hipGetLastError());
    }
    if (lanes > 1) PANDA_TRY(hipStreamWaitEvent(stream, phase_events.fixed[nranges - 1], 0));
    PANDA_TRY(mark(5));
    {
        u32 *d_rows = d_l1, *d_cols = d_l1 + (size_t)lists * rc_rows * PW;
        hipLaunchKernelGGL(k_sum_lines<Fq>, dim3(rc_rows + rc_cols * rc_csplit, lists), dim3(64), 0, stream, d_bacc, d_rows, d_cols, rc_rows, rc_cols, rc_csplit);
        const unsigned emit = tabled ? (cfg.msm_result_coordinate_type == PROJECTIVE ? 2u : 1u) : 0u;
        hipLaunchKernelGGL(k_weighted_slots<Fq>, dim3(slots, lists, rc_parts), dim3(64), 0, stream, d_rows, d_cols, d_slots, rc_a, rc_b, rc_csplit);
        hipLaunchKernelGGL(k_slot_total<Fq>, dim3(lists), dim3(64), 0, stream, d_slots, res_dev ? res_dev : h_win, slots * rc_parts, emit);
    }
    PANDA_TRY(mark(6));
    PANDA_TRY(hipGetLastError());
    if (res_dev) PANDA_TRY(mark(7)); // nothing follows on the device
    PANDA_TRY(hipStreamSynchronize(stream));
    if (*h_flag != 0) { // the caller's buffer is not what was registered: the sums mean nothing (and k_accumulate skipped its work)
        if (stale) *stale = true;
        return hipSuccess;
    }
    if (!res_dev) {
        if (tabled) // the result itself came through the mailbox
            PANDA_TRY(hipMemcpyAsync(cfg.results, h_win, 3 * LQ * 4, hipMemcpyDefault, stream));
        else {
            std::vector<Xyzz<Fq>> windows(lists);
            for (unsigned w = 0; w < lists; w++) {
                const u32 *src = h_win + (size_t)w * PW;
                for (int i = 0; i < Fq::N; i++) {
                    windows[w].X.l[i] = src[i];
                    windows[w].Y.l[i] = src[Fq::N + i];
                    windows[w].ZZ.l[i] = src[2 * Fq::N + i];
                    windows[w].ZZZ.l[i] = src[3 * Fq::N + i];
                }
            }
            Xyzz<Fq> result;
            host_horner(result, windows, plan);
            u32 out[3 * LQ];
            if (cfg.msm_result_coordinate_type == PROJECTIVE)
                xyzz_to_homogeneous_wire(out, result);
            else
                xyzz_to_jacobian_wire(out, result);
            // results may be a device pointer (unit.rs:32-47) or pinned host memory (msm_test.cu:53,125)
            PANDA_TRY(hipMemcpyAsync(cfg.results, out, sizeof(out), hipMemcpyDefault, stream));
        }
        PANDA_TRY(mark(7));
        PANDA_TRY(hipStreamSynchronize(stream));
    }

    if (stamps) panda::thread_msm_clock() = panda::clock_delta(stamp_block, stamp_block + 2 * panda::CLOCK_STAMP_SLOTS);
    float ms = 0;
    for (int i = 0; i < 6; i++) { // phases whose events were not recorded in this call read 0
        phase_ms[i] = 0;
        if (wanted(i) && wanted(i + 1) && hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) phase_ms[i] = ms;
    }
    phase_ms[6] = phase_ms[7] = 0;
    if (wanted(6) && wanted(7) && hipEventElapsedTime(&ms, ev[6], ev[7]) == hipSuccess) phase_ms[6] = ms;
    if (wanted(0) && wanted(6) && hipEventElapsedTime(&ms, ev[0], ev[6]) == hipSuccess) phase_ms[7] = ms;
    return hipSuccess;
}

template <class Fq>
hipError_t build_registration(panda::MsmRegistration &r, hipStream_t s)
{
    const u64 n = (u64)1 << r.log_n;
    const size_t row = 2 * Fq::L * 4;
    const unsigned tables = r.tabled ? r.plan.W : 1u;
    r.bytes = (size_t)tables * n * row;
    const size_t tail = panda::align256(r.bytes);
    PANDA_TRY(hipMalloc(&r.converted, tail + panda::REG_SAMPLES * row));
    u32 *t0 = (u32 *)r.converted;
    u32 *samples = (u32 *)((char *)r.converted + tail);
    r.samples = samples;
    hipLaunchKernelGGL(k_take_samples, dim3(1), dim3(panda::REG_SAMPLES * 16), 0, s, (const u32 *)r.wire, samples, n, 2u * Fq::L);
    hipLaunchKernelGGL(k_convert_bases<Fq>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const u32 *)r.wire, t0, n);
    for (unsigned k = 1; k < tables; k++)
        hipLaunchKernelGGL(k_table_step<Fq>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, t0 + (size_t)(k - 1) * n * 2 * Fq::L,
                           t0 + (size_t)k * n * 2 * Fq::L, n, (unsigned)r.plan.width[k - 1]);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = panda::msm_hash_wire(r.wire, (size_t)n * row, s, &r.hash); // synchronises
    if (e != hipSuccess) {
        (void)hipFree(r.converted);
        r.converted = nullptr;
    }
    return e;
}

} // namespace

#endif // PANDA_MSM_IMPL
