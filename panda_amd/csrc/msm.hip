// msm.hip -- Pippenger multi-scalar multiplication for gfx950.
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
//   each bucket weighted by c doublings + adds         segmented running sums (2 adds per bucket) + one short
//     (:411-420, ~240 mulmods per bucket)                double-and-add per 4 buckets + tree reduction
//   Jacobian madd 7M+4S on 8x32-bit PTX carry chains   XYZZ madd 8M+2S on 9x29-bit limbs / v_mad_u64_u32 (fe29.h)
//   9 cudaDeviceSynchronize per call (:611-755)        one stream, one synchronisation before the host Horner
//   scratch cudaMallocAsync'd and freed per call       per-thread arena kept between calls
//
// Pipeline (all on cfg.stream):
//   k_convert_bases   wire affine -> internal Montgomery radix, 64 B/point (96 B BLS12-377)
//   k_digits          scalar -> canonical -> W signed c-bit digits (u16 codes, window-major)
//   k_part_hist/scan/scatter  level 1 of the sort: (id, sign, lo bits) words into 2^hi partitions, LDS histograms and cursors
//   k_bucket_sort     level 2: one workgroup per partition ranks the lo bits in LDS -> bucket offsets + point-id lists
//   k_accumulate      flat chunks of K entries: acc += +/- base   (the hot kernel)
//   k_fixup           merge bucket pieces that straddle chunks
//   k_reduce_groups   4 buckets -> one weighted partial
//   k_tree_reduce     partials -> one point per window
//   host              Horner over the W window sums (as the reference does, msm_cuda.cuh:738-743), output conversion
#include <algorithm>
#include <mutex>
#include <vector>

#include "curve29.h"
#include "panda_internal.h"

using namespace panda29;

namespace panda {

// ---------------------------------------------------------------- arena
hipError_t Arena::reserve(size_t bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (base && (dev != device || bytes > capacity)) {
        e = hipDeviceSynchronize();
        if (e != hipSuccess) return e;
        e = hipFree(base);
        if (e != hipSuccess) return e;
        base = nullptr;
        capacity = 0;
    }
    if (!base) {
        size_t want = bytes + (bytes >> 3) + (1u << 20);
        e = hipMalloc(&base, want);
        if (e != hipSuccess) {
            base = nullptr;
            return e;
        }
        capacity = want;
        device = dev;
    }
    used = 0;
    return hipSuccess;
}

hipError_t Arena::release()
{
    hipError_t e = hipSuccess;
    if (base) {
        (void)hipDeviceSynchronize();
        e = hipFree(base);
    }
    base = nullptr;
    capacity = used = 0;
    device = -1;
    return e;
}

Arena &thread_arena()
{
    static thread_local Arena arena;
    return arena;
}

hipError_t release_thread_arena() { return thread_arena().release(); }

} // namespace panda

namespace {

constexpr u32 DIGIT_ZERO = 0x7fffu; // "+2^15" cannot occur with the recoding below, so it encodes digit 0
constexpr int GROUP = 4;            // buckets per k_reduce_groups thread

struct CurveBn254 {
    typedef Bn254Fq Fq;
    typedef Bn254Fr Fr;
};
struct CurveBls377 {
    typedef Bls377Fq Fq;
    typedef Bls377Fr Fr;
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

// scalar (Montgomery wire form) -> W signed digits.  code = (neg << 15) | (|d| - 1), DIGIT_ZERO for d = 0.
// Replaces init_handle_scalars_kernel + the slice extraction of calc_lens/fill_arrs (msm_cuda.cuh:148-205,232-282).
// Window layout: BITS+1 scalar bits (one spare for the signed-digit carry) cut into W windows whose widths differ by
// at most one, so that the top window is as wide as the others (a narrow top window would put all n points into a
// handful of buckets).  width[k] <= 16.
struct WindowPlan {
    unsigned W;
    unsigned char width[64];
    unsigned short lo[64];
};

template <class Fr>
__global__ void __launch_bounds__(256) k_digits(const u32 *__restrict__ scalars, uint16_t *__restrict__ dig, u64 n, WindowPlan plan)
{
    constexpr int L = Fr::L;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[L], s[L + 1];
    load_words<L>(w, scalars + i * L);
    fe_wire_to_canonical<Fr>(s, w);
    s[L] = 0;
    u32 carry = 0;
    for (unsigned k = 0; k < plan.W; k++) {
        const unsigned c = plan.width[k];
        const u32 half = 1u << (c - 1), full = 1u << c, mask = full - 1;
        unsigned lo = plan.lo[k], m = lo >> 5, sh = lo & 31;
        u32 raw = 0;
        if (m < (unsigned)L) {
            u64 v = s[m] | ((u64)s[m + 1] << 32);
            raw = (u32)(v >> sh) & mask;
        }
        raw += carry;
        u32 code;
        if (raw >= half) { // negative digit raw - 2^c (or zero when raw == 2^c)
            u32 mag = full - raw;
            carry = 1;
            code = mag ? (0x8000u | (mag - 1)) : DIGIT_ZERO;
        } else {
            carry = 0;
            code = raw ? (raw - 1) : DIGIT_ZERO;
        }
        dig[(u64)k * n + i] = (uint16_t)code;
    }
}

// ---- (window, bucket) -> point-id lists: two-level partition sort staged in LDS --------------------------------
// Replaces the reference's three global-atomic passes (calc_lens / allo_arrs / fill_arrs, msm_cuda.cuh:159-282).
// Bucket ids are split into `hi` (partition) and `lo` bits.  Level 1 moves every (id, sign, lo) word into its
// partition with per-tile LDS histograms and LDS cursors; level 2 gives each partition to one workgroup, which
// counts and ranks its `lo` values in LDS and writes the final order.  A partition's output range equals its
// input range, so no global prefix over the 2^(c-1) buckets is needed.  No global atomics anywhere.
constexpr unsigned SORT_TILE = 8192; // digits per workgroup in level 1
constexpr unsigned MAX_PARTS = 1024;

struct SortGeom {
    unsigned log_n, lo_bits, H, tiles;
};

__global__ void __launch_bounds__(256) k_part_hist(const uint16_t *__restrict__ dig, u32 *__restrict__ tile_hist, SortGeom g)
{
    __shared__ u32 h[MAX_PARTS];
    const unsigned w = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    for (unsigned i = tid; i < g.H; i += 256) h[i] = 0;
    __syncthreads();
    const u64 n = (u64)1 << g.log_n;
    const uint16_t *dw = dig + ((u64)w << g.log_n);
    const u64 begin = (u64)tile * SORT_TILE, end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
    for (u64 i = begin + tid; i < end; i += 256) {
        u32 code = dw[i];
        if (code != DIGIT_ZERO) atomicAdd(&h[(code & 0x7fffu) >> g.lo_bits], 1u);
    }
    __syncthreads();
    u32 *out = tile_hist + ((u64)w * g.tiles + tile) * g.H;
    for (unsigned i = tid; i < g.H; i += 256) out[i] = h[i];
}

// tile_hist -> exclusive prefix over tiles (in place), one WAVE per (window, partition) column: 64 tiles per step with a
// shuffle scan instead of one dependent load per tile; column totals go to `totals`
__global__ void __launch_bounds__(1024) k_part_scan_cols(u32 *__restrict__ tile_hist, u32 *__restrict__ totals, SortGeom g)
{
    const unsigned w = blockIdx.y, lane = threadIdx.x & 63, h = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (h >= g.H) return; // whole wave exits together
    u32 *col = tile_hist + (u64)w * g.tiles * g.H + h;
    u32 run = 0;
    for (unsigned t0 = 0; t0 < g.tiles; t0 += 64) {
        const unsigned tile = t0 + lane;
        u32 v = tile < g.tiles ? col[(u64)tile * g.H] : 0;
        u32 inc = v;
        for (unsigned d = 1; d < 64; d <<= 1) {
            u32 up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        if (tile < g.tiles) col[(u64)tile * g.H] = run + inc - v;
        run += __shfl(inc, 63, 64);
    }
    if (lane == 0) totals[(u64)w * g.H + h] = run;
}

// one block per window: exclusive scan of the H column totals -> part_off[w][0..H]
__global__ void __launch_bounds__(1024) k_part_offsets(const u32 *__restrict__ totals, u32 *__restrict__ part_off, SortGeom g)
{
    __shared__ u32 tot[MAX_PARTS];
    const unsigned w = blockIdx.x, t = threadIdx.x;
    const u32 mine = t < g.H ? totals[(u64)w * g.H + t] : 0;
    tot[t] = mine;
    __syncthreads();
    for (unsigned d = 1; d < 1024; d <<= 1) {
        u32 v = (t >= d) ? tot[t - d] : 0;
        __syncthreads();
        tot[t] += v;
        __syncthreads();
    }
    if (t < g.H) part_off[(u64)w * (g.H + 1) + t] = tot[t] - mine;
    if (t == 1023) part_off[(u64)w * (g.H + 1) + g.H] = tot[1023];
}

constexpr unsigned SORT_THREADS = 1024; // 16 waves per workgroup: these kernels wait on LDS atomics and HBM, they need the occupancy

// block-wide exclusive scan of `count` (<= 1024) LDS words in place by SORT_THREADS threads
__device__ __forceinline__ void block_exclusive_scan(u32 *a, unsigned count, u32 *scratch /* SORT_THREADS words */)
{
    const unsigned tid = threadIdx.x;
    const unsigned per = (count + SORT_THREADS - 1) / SORT_THREADS;
    u32 local = 0;
    for (unsigned j = 0; j < per; j++) {
        unsigned i = tid * per + j;
        if (i < count) local += a[i];
    }
    scratch[tid] = local;
    __syncthreads();
    for (unsigned d = 1; d < SORT_THREADS; d <<= 1) {
        u32 v = (tid >= d) ? scratch[tid - d] : 0;
        __syncthreads();
        scratch[tid] += v;
        __syncthreads();
    }
    u32 run = scratch[tid] - local;
    for (unsigned j = 0; j < per; j++) {
        unsigned i = tid * per + j;
        if (i < count) {
            u32 v = a[i];
            a[i] = run;
            run += v;
        }
    }
    __syncthreads();
}

// word written to the partition buffer: [lo : lo_bits][sign : 1][point id : log_n].
// The tile is first grouped by partition in LDS (local counting sort), then written out linearly, so that a wave
// stores runs of consecutive addresses instead of 64 unrelated 4-byte words.
__global__ void __launch_bounds__(SORT_THREADS) k_part_scatter(const uint16_t *__restrict__ dig, const u32 *__restrict__ tile_hist, const u32 *__restrict__ part_off,
                                                      u32 *__restrict__ p1, SortGeom g)
{
    __shared__ u32 lstart[MAX_PARTS]; // local start of each partition's run in the staging buffer
    __shared__ u32 lcur[MAX_PARTS];   // local cursor
    __shared__ u32 gbase[MAX_PARTS];  // global position of this tile's first element of the partition
    __shared__ u32 words[SORT_TILE];
    __shared__ uint16_t parts_of[SORT_TILE];
    __shared__ u32 scratch[SORT_THREADS];
    const unsigned w = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const u32 *base = tile_hist + ((u64)w * g.tiles + tile) * g.H;
    const u32 *po = part_off + (u64)w * (g.H + 1);
    for (unsigned i = tid; i < g.H; i += SORT_THREADS) {
        lstart[i] = 0;
        lcur[i] = 0;
        gbase[i] = po[i] + base[i];
    }
    __syncthreads();
    const u64 n = (u64)1 << g.log_n;
    const uint16_t *dw = dig + ((u64)w << g.log_n);
    u32 *pw = p1 + ((u64)w << g.log_n);
    const u64 begin = (u64)tile * SORT_TILE, end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
    const u32 lo_mask = (1u << g.lo_bits) - 1;
    for (u64 i = begin + tid; i < end; i += SORT_THREADS) {
        u32 code = dw[i];
        if (code != DIGIT_ZERO) atomicAdd(&lstart[(code & 0x7fffu) >> g.lo_bits], 1u);
    }
    __syncthreads();
    block_exclusive_scan(lstart, g.H, scratch);
    for (u64 i = begin + tid; i < end; i += SORT_THREADS) {
        u32 code = dw[i];
        if (code == DIGIT_ZERO) continue;
        u32 b = code & 0x7fffu, h = b >> g.lo_bits;
        u32 slot = lstart[h] + atomicAdd(&lcur[h], 1u);
        words[slot] = ((b & lo_mask) << (g.log_n + 1)) | ((code >> 15) << g.log_n) | (u32)i;
        parts_of[slot] = (uint16_t)h;
    }
    __syncthreads();
    const u32 total = lstart[g.H - 1] + lcur[g.H - 1];
    for (u32 j = tid; j < total; j += SORT_THREADS) {
        u32 h = parts_of[j];
        pw[gbase[h] + (j - lstart[h])] = words[j];
    }
}

// one workgroup per (partition, window): count the lo values, publish the bucket offsets, then rank the ids chunk by
// chunk in LDS and write each chunk out as runs
constexpr unsigned BS_CHUNK = 8192;

__global__ void __launch_bounds__(SORT_THREADS) k_bucket_sort(const u32 *__restrict__ p1, const u32 *__restrict__ part_off, u32 *__restrict__ off,
                                                     u32 *__restrict__ sorted, SortGeom g, unsigned NB)
{
    __shared__ u32 cnt[128], cur[128], lstart[128], lcur[128];
    __shared__ u32 words[BS_CHUNK];
    __shared__ unsigned char lo_of[BS_CHUNK];
    const unsigned w = blockIdx.y, h = blockIdx.x, tid = threadIdx.x;
    const unsigned L = 1u << g.lo_bits;
    const u32 ps = part_off[(u64)w * (g.H + 1) + h], pe = part_off[(u64)w * (g.H + 1) + h + 1];
    const u32 *pw = p1 + ((u64)w << g.log_n);
    u32 *sw = sorted + ((u64)w << g.log_n);
    const unsigned shift = g.log_n + 1;
    if (tid < 128) cnt[tid] = 0;
    __syncthreads();
    for (u32 j = ps + tid; j < pe; j += SORT_THREADS) atomicAdd(&cnt[pw[j] >> shift], 1u);
    __syncthreads();
    u32 mine = tid < 128 ? cnt[tid] : 0;
    for (unsigned d = 1; d < 128; d <<= 1) { // inclusive scan of the (at most 128) counts
        u32 v = (tid < 128 && tid >= d) ? cnt[tid - d] : 0;
        __syncthreads();
        if (tid < 128) cnt[tid] += v;
        __syncthreads();
    }
    if (tid < L) {
        u32 start = ps + cnt[tid] - mine;
        cur[tid] = start;
        off[(u64)w * (NB + 1) + ((u64)h << g.lo_bits) + tid] = start;
    }
    if (h == g.H - 1 && tid == 0) off[(u64)w * (NB + 1) + NB] = pe;
    __syncthreads();
    const u32 id_mask = (1u << g.log_n) - 1;
    for (u32 cbeg = ps; cbeg < pe; cbeg += BS_CHUNK) { // pe, ps are uniform over the block: barriers are safe
        const u32 cend = min(cbeg + BS_CHUNK, pe);
        if (tid < 128) {
            lstart[tid] = 0;
            lcur[tid] = 0;
        }
        __syncthreads();
        for (u32 j = cbeg + tid; j < cend; j += SORT_THREADS) atomicAdd(&lstart[pw[j] >> shift], 1u);
        __syncthreads();
        u32 c0 = tid < 128 ? lstart[tid] : 0;
        for (unsigned d = 1; d < 128; d <<= 1) {
            u32 v = (tid < 128 && tid >= d) ? lstart[tid - d] : 0;
            __syncthreads();
            if (tid < 128) lstart[tid] += v;
            __syncthreads();
        }
        if (tid < 128) {
            lstart[tid] -= c0; // exclusive
            cnt[tid] = c0;
        }
        __syncthreads();
        for (u32 j = cbeg + tid; j < cend; j += SORT_THREADS) {
            u32 v = pw[j];
            u32 l = v >> shift;
            u32 slot = lstart[l] + atomicAdd(&lcur[l], 1u);
            words[slot] = (v & id_mask) | (((v >> g.log_n) & 1u) << 31);
            lo_of[slot] = (unsigned char)l;
        }
        __syncthreads();
        for (u32 j = tid; j < cend - cbeg; j += SORT_THREADS) {
            u32 l = lo_of[j];
            sw[cur[l] + (j - lstart[l])] = words[j];
        }
        __syncthreads();
        if (tid < 128) cur[tid] += cnt[tid];
        __syncthreads();
    }
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

// a converted base as it sits in HBM: 2*L words, all zero for the identity
template <class F>
struct PackedBase {
    u32 w[2 * F::L];
};

template <class F>
__device__ __forceinline__ void fetch_base(PackedBase<F> &b, const u32 *bases, u32 entry)
{
    load_words<2 * F::L>(b.w, bases + (u64)(entry & 0x7fffffffu) * 2 * F::L);
}

template <class F>
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
        fe_neg<F, 1>(ny, y); // y is canonical (< p)
        y = ny;
    }
}

// The hot kernel.  Thread t of window w owns sorted entries [t*K, t*K + K) of that window and adds the
// bases they name into the accumulators of the buckets they fall in; the first and last bucket of a
// chunk may continue in the neighbouring chunks, those pieces go to `parts` and are merged by k_fixup.
// Replaces aggerate_buckets_groups_kernel's per-bucket list walk (msm_cuda.cuh:373-409).
template <class F>
__global__ void __launch_bounds__(128, (F::N <= 9 ? 4 : 2)) k_accumulate(const u32 *__restrict__ bases, const u32 *__restrict__ sorted, const u32 *__restrict__ off,
                                                    u32 *__restrict__ bucket_acc, u32 *__restrict__ parts, unsigned log_n, unsigned NB, unsigned K,
                                                    unsigned chunks)
{
    constexpr int PW = 4 * F::N;
    const unsigned w = blockIdx.y;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= chunks) return;
    const u32 *ow = off + (u64)w * (NB + 1);
    const u32 nw = ow[NB];
    const u32 start = t * K;
    if (start >= nw) return;
    const u32 end = min(start + K, nw);
    const u32 *sw = sorted + ((u64)w << log_n);
    u32 *pw = parts + ((u64)w * chunks + t) * 2 * PW;
    u32 *bw = bucket_acc + (u64)w * NB * PW;

    u32 b = owner_bucket(ow, NB, start);
    u32 next = ow[b + 1];
    Xyzz<F> acc;
    xyzz_set_identity(acc);

    PackedBase<F> next_base;
    u32 next_entry = sw[start];
    fetch_base<F>(next_base, bases, next_entry);
    for (u32 pos = start; pos < end; pos++) {
        if (pos >= next) { // the run of bucket b ends here
            const bool complete = ow[b] >= start; // its end (== pos) is inside the chunk by construction
            store_xyzz<F>(complete ? bw + (u64)b * PW : pw, acc);
            xyzz_set_identity(acc);
            do {
                b++;
                next = ow[b + 1];
            } while (next <= pos);
        }
        Fe<F> cx, cy;
        bool cinf;
        const u32 entry = next_entry;
        unpack_base<F>(cx, cy, cinf, next_base, entry);
        if (pos + 1 < end) { // prefetch the next base (still packed: 16 registers) under this addition
            next_entry = sw[pos + 1];
            fetch_base<F>(next_base, bases, next_entry);
        }
        if (cinf) continue;
        if (xyzz_is_identity(acc)) {
            xyzz_from_affine(acc, cx, cy);
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
    const bool starts_inside = ow[b] >= start;
    const bool ends_inside = next <= end;
    u32 *dst = (starts_inside && ends_inside) ? bw + (u64)b * PW : (starts_inside ? pw + PW : pw);
    store_xyzz<F>(dst, acc);
}

// bucket pieces: a bucket that spans chunks t0 < t1 is the LAST run of t0 (stored in slot 1, or slot 0 if it
// also is t0's first run and started earlier -- impossible here since t0 = start / K), the ONLY run of every
// chunk strictly between (slot 0) and the FIRST run of t1 (slot 0).
// Buckets cut into more than LONG_SPAN pieces (heavily skewed scalars) are queued for k_fixup_long instead of being
// summed by one thread.
constexpr unsigned LONG_SPAN = 128;
constexpr unsigned LONG_BLOCKS = 256; // workgroups per window that serve the queue

template <class F>
__global__ void __launch_bounds__(128) k_fixup(const u32 *__restrict__ off, const u32 *__restrict__ parts, u32 *__restrict__ bucket_acc, unsigned NB,
                                               unsigned K, unsigned chunks, u32 *__restrict__ long_count, u32 *__restrict__ long_list, unsigned long_cap)
{
    constexpr int PW = 4 * F::N;
    const unsigned w = blockIdx.y;
    const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= NB) return;
    const u32 *ow = off + (u64)w * (NB + 1);
    const u32 s = ow[b], e = ow[b + 1];
    if (s == e) return; // empty: bucket_acc was zeroed (identity)
    const u32 t0 = s / K, t1 = (e - 1) / K;
    if (t0 == t1) return; // lies inside one chunk: written by k_accumulate
    if (t1 - t0 > LONG_SPAN) {
        u32 slot = atomicAdd(&long_count[w], 1u); // a handful per window at most: (t1 - t0) > LONG_SPAN bounds it by chunks / LONG_SPAN
        if (slot < long_cap) {
            u32 *e3 = long_list + ((u64)w * long_cap + slot) * 3;
            e3[0] = b;
            e3[1] = t0;
            e3[2] = t1;
        }
        return;
    }
    const u32 *pw = parts + (u64)w * chunks * 2 * PW;
    Xyzz<F> acc, q;
    load_xyzz<F>(acc, pw + ((u64)t0 * 2 + 1) * PW);
    for (u32 t = t0 + 1; t <= t1; t++) {
        load_xyzz<F>(q, pw + (u64)t * 2 * PW);
        xyzz_add(acc, q);
    }
    store_xyzz<F>(bucket_acc + ((u64)w * NB + b) * PW, acc);
}

// one workgroup per queued bucket: 256 threads stride over its pieces, then an LDS tree
template <class F>
__global__ void __launch_bounds__(256) k_fixup_long(const u32 *__restrict__ parts, u32 *__restrict__ bucket_acc, unsigned NB, unsigned chunks,
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
            xyzz_add(acc, q);
        }
        store_xyzz<F>(lds + t * PW, acc);
        __syncthreads();
        for (unsigned s = 128; s > 0; s >>= 1) {
            if (t < s) {
                load_xyzz<F>(q, lds + (t + s) * PW);
                xyzz_add(acc, q);
                store_xyzz<F>(lds + t * PW, acc);
            }
            __syncthreads();
        }
        if (t == 0) store_xyzz<F>(bucket_acc + ((u64)w * NB + b) * PW, acc);
        __syncthreads();
    }
}

// k * p for a small scalar k (double-and-add from the top bit)
template <class F>
__device__ __forceinline__ void xyzz_mul_small(Xyzz<F> &r, const Xyzz<F> &p, u32 k)
{
    Xyzz<F> acc, d;
    xyzz_set_identity(acc);
    if (k == 0 || xyzz_is_identity(p)) {
        r = acc;
        return;
    }
    int top = 31 - __clz(k);
    for (int bit = top; bit >= 0; bit--) {
        xyzz_dbl(d, acc);
        acc = d;
        if ((k >> bit) & 1) xyzz_add(acc, p);
    }
    r = acc;
}

// thread (w, g): buckets b = g*GROUP .. g*GROUP+GROUP-1, weights b+1.
//   running sums from the top give S = sum B_j and T = sum (j+1) B_j (j local); the partial is T + (g*GROUP) * S.
// Replaces the per-bucket c-step double-and-add of msm_cuda.cuh:411-420.
template <class F>
__global__ void __launch_bounds__(128) k_reduce_groups(const u32 *__restrict__ bucket_acc, u32 *__restrict__ out, unsigned NB, unsigned groups)
{
    constexpr int PW = 4 * F::N;
    const unsigned w = blockIdx.y;
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    const u32 *bw = bucket_acc + (u64)w * NB * PW;
    Xyzz<F> run, sum, q;
    xyzz_set_identity(run);
    xyzz_set_identity(sum);
    for (int j = GROUP - 1; j >= 0; j--) {
        u32 b = g * GROUP + j;
        if (b < NB) {
            load_xyzz<F>(q, bw + (u64)b * PW);
            xyzz_add(run, q);
        }
        xyzz_add(sum, run);
    }
    xyzz_mul_small(q, run, g * GROUP);
    xyzz_add(sum, q);
    store_xyzz<F>(out + ((u64)w * groups + g) * PW, sum);
}

// block b of window w sums in[w][b*per .. b*per+per) -> out[w][b]
template <class F>
__global__ void __launch_bounds__(256) k_tree_reduce(const u32 *__restrict__ in, u32 *__restrict__ out, unsigned count, unsigned per)
{
    constexpr int PW = 4 * F::N;
    __shared__ __attribute__((aligned(16))) u32 lds[256 * PW];
    const unsigned w = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
    const unsigned begin = blk * per, end = min(begin + per, count);
    Xyzz<F> acc, q;
    xyzz_set_identity(acc);
    for (unsigned i = begin + t; i < end; i += 256) {
        load_xyzz<F>(q, in + ((u64)w * count + i) * PW);
        xyzz_add(acc, q);
    }
    store_xyzz<F>(lds + t * PW, acc);
    __syncthreads();
    for (unsigned s = 128; s > 0; s >>= 1) {
        if (t < s) {
            load_xyzz<F>(q, lds + (t + s) * PW);
            xyzz_add(acc, q);
            store_xyzz<F>(lds + t * PW, acc);
        }
        __syncthreads();
    }
    if (t == 0) store_xyzz<F>(out + ((u64)w * gridDim.x + blk) * PW, acc);
}

// ------------------------------------------------------------------------------- host side

thread_local float g_phase_ms[PANDA_MSM_PHASES] = {0};

// Cached bases (README "Supports cached bases and scalars"; init_msm, wrapper.rs:122-152): a caller that keeps a base
// set on the device for many MSMs may register it; the radix conversion k_convert_bases would repeat on every call is
// then done once and kept next to it.  The caller promises not to modify a registered buffer until it is unregistered.
struct RegisteredBases {
    const void *wire; // the caller's device pointer (the key)
    void *converted;
    unsigned log_n, curve;
    int device;
};
std::mutex g_registry_mutex;
std::vector<RegisteredBases> g_registry;

const void *lookup_registered(const void *wire, unsigned log_n, unsigned curve)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    for (const auto &r : g_registry)
        if (r.wire == wire && r.log_n == log_n && r.curve == curve && r.device == dev) return r.converted;
    return nullptr;
}
unsigned g_window_override = 0;

const char *const kPhaseNames[PANDA_MSM_PHASES] = {"convert_bases+digits", "sort_partition", "sort_buckets", "accumulate",
                                                   "fixup", "bucket_reduce", "d2h+host_horner", "total_device"};

// window width policy (replaces get_window_bits_count, msm_cuda.cuh:21-45)
unsigned pick_window_bits(unsigned log_n)
{
    if (g_window_override) return std::min(std::max(g_window_override, 2u), 16u);
    int c = (int)log_n - 4;
    return (unsigned)std::min(std::max(c, 4), 16);
}

WindowPlan make_plan(unsigned total_bits, unsigned c)
{
    WindowPlan p{};
    p.W = (total_bits + c - 1) / c;
    const unsigned base = total_bits / p.W, rem = total_bits % p.W;
    unsigned lo = 0;
    for (unsigned k = 0; k < p.W; k++) {
        p.width[k] = (unsigned char)(base + (k < rem ? 1 : 0));
        p.lo[k] = (unsigned short)lo;
        lo += p.width[k];
    }
    return p;
}

template <class F>
void host_horner(Xyzz<F> &result, const std::vector<Xyzz<F>> &windows, const WindowPlan &plan)
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

template <class C>
hipError_t msm_execute(const panda_msm_configuration &cfg)
{
    typedef typename C::Fq Fq;
    typedef typename C::Fr Fr;
    constexpr int PW = 4 * Fq::N;
    constexpr int LQ = Fq::L, LR = Fr::L;
    hipStream_t stream = static_cast<hipStream_t>(cfg.stream.handle);
    const unsigned log_n = cfg.log_scalars_count;
    if (log_n > 26 || !cfg.bases || !cfg.scalars || !cfg.results) return hipErrorInvalidValue;
    const u64 n = (u64)1 << log_n;
    // BITS + 1: one spare bit for the signed-digit carry.  The top window must never carry out: its largest raw value
    // (top bits of r - 1, plus a carry in) has to stay below half its range; widen the plan by a bit until it does.
    WindowPlan plan{};
    for (unsigned total = Fr::BITS + 1;; total++) {
        plan = make_plan(total, pick_window_bits(log_n));
        const unsigned lo = plan.lo[plan.W - 1], m = lo >> 5, sh = lo & 31;
        u64 top = m < (unsigned)LR ? ((u64)Fr::PW[m] >> sh) : 0;
        if (m + 1 < (unsigned)LR) top |= (u64)Fr::PW[m + 1] << (32 - sh);
        if (m + 2 < (unsigned)LR && sh) top |= (u64)Fr::PW[m + 2] << (64 - sh);
        if (top + 1 < ((u64)1 << (plan.width[plan.W - 1] - 1)) || total > Fr::BITS + 8) break;
    }
    const unsigned W = plan.W;
    const unsigned c = plan.width[0]; // widest window
    const unsigned NB = 1u << (c - 1);
    const unsigned K = log_n >= 24 ? 128 : (log_n >= 22 ? 64 : (log_n >= 16 ? 32 : 16)); // sorted entries per accumulate thread
    const unsigned chunks = (unsigned)((n + K - 1) / K);
    const unsigned groups = (NB + GROUP - 1) / GROUP;
    const unsigned lvl1 = (groups + 255) / 256 > 64 ? 64 : (groups + 255) / 256; // blocks in the first tree level
    const unsigned per1 = (groups + lvl1 - 1) / lvl1;

    // ---- scratch
    const size_t sz_bases = panda::align256(n * 2 * LQ * 4);
    const size_t sz_dig = panda::align256(n * W * 2);
    SortGeom geom;
    geom.log_n = log_n;
    geom.lo_bits = std::min(std::min(7u, c - 1), 31u - log_n);
    geom.H = 1u << (c - 1 - geom.lo_bits);
    geom.tiles = (unsigned)((n + SORT_TILE - 1) / SORT_TILE);
    if (geom.H > MAX_PARTS) return hipErrorInvalidValue;
    const size_t sz_thist = panda::align256((size_t)W * geom.tiles * geom.H * 4);
    const size_t sz_poff = panda::align256((size_t)W * (geom.H + 1) * 4);
    const size_t sz_off = panda::align256((size_t)W * (NB + 1) * 4);
    const size_t sz_sorted = panda::align256(n * W * 4);
    const size_t sz_bacc = panda::align256((size_t)W * NB * PW * 4);
    const size_t sz_parts = panda::align256((size_t)W * chunks * 2 * PW * 4);
    const size_t sz_gsum = panda::align256((size_t)W * groups * PW * 4);
    const size_t sz_l1 = panda::align256((size_t)W * lvl1 * PW * 4);
    const size_t sz_win = panda::align256((size_t)W * PW * 4);
    const unsigned long_cap = chunks / LONG_SPAN + 2;
    const size_t sz_lcount = panda::align256((size_t)W * 4);
    const size_t sz_llist = panda::align256((size_t)W * long_cap * 3 * 4);
    panda::Arena &arena = panda::thread_arena();
    PANDA_TRY(arena.reserve(sz_bases + sz_dig + sz_thist + 2 * sz_poff + sz_off + 2 * sz_sorted + sz_bacc + sz_parts + sz_gsum + sz_l1 + sz_win + sz_lcount + sz_llist + 4096));
    u32 *d_bases = (u32 *)arena.take(sz_bases);
    uint16_t *d_dig = (uint16_t *)arena.take(sz_dig);
    u32 *d_thist = (u32 *)arena.take(sz_thist);
    u32 *d_poff = (u32 *)arena.take(sz_poff);
    u32 *d_ptot = (u32 *)arena.take(sz_poff);
    u32 *d_p1 = (u32 *)arena.take(sz_sorted);
    u32 *d_off = (u32 *)arena.take(sz_off);
    u32 *d_sorted = (u32 *)arena.take(sz_sorted);
    u32 *d_bacc = (u32 *)arena.take(sz_bacc);
    u32 *d_parts = (u32 *)arena.take(sz_parts);
    u32 *d_gsum = (u32 *)arena.take(sz_gsum);
    u32 *d_l1 = (u32 *)arena.take(sz_l1);
    u32 *d_win = (u32 *)arena.take(sz_win);
    u32 *d_lcount = (u32 *)arena.take(sz_lcount);
    u32 *d_llist = (u32 *)arena.take(sz_llist);
    if (!d_win || !d_llist) return hipErrorOutOfMemory;

    struct PhaseEvents { // destroyed on every exit path
        hipEvent_t ev[8] = {};
        ~PhaseEvents()
        {
            for (auto &e : ev)
                if (e) (void)hipEventDestroy(e);
        }
    } phase_events;
    hipEvent_t(&ev)[8] = phase_events.ev;
    for (auto &e : ev) PANDA_TRY(hipEventCreate(&e));
    auto mark = [&](int i) { return hipEventRecord(ev[i], stream); };

    PANDA_TRY(mark(0));
    const unsigned blocks_n = (unsigned)((n + 255) / 256);
    const u32 *registered = (const u32 *)lookup_registered(cfg.bases, log_n, Fq::N == 9 ? 0u : 1u);
    if (registered)
        d_bases = const_cast<u32 *>(registered); // converted once at registration
    else
        hipLaunchKernelGGL(k_convert_bases<Fq>, dim3(blocks_n), dim3(256), 0, stream, (const u32 *)cfg.bases, d_bases, n);
    hipLaunchKernelGGL(k_digits<Fr>, dim3(blocks_n), dim3(256), 0, stream, (const u32 *)cfg.scalars, d_dig, n, plan);
    PANDA_TRY(mark(1));
    hipLaunchKernelGGL(k_part_hist, dim3(geom.tiles, W), dim3(256), 0, stream, d_dig, d_thist, geom);
    hipLaunchKernelGGL(k_part_scan_cols, dim3((geom.H + 15) / 16, W), dim3(1024), 0, stream, d_thist, d_ptot, geom);
    hipLaunchKernelGGL(k_part_offsets, dim3(W), dim3(1024), 0, stream, d_ptot, d_poff, geom);
    hipLaunchKernelGGL(k_part_scatter, dim3(geom.tiles, W), dim3(SORT_THREADS), 0, stream, d_dig, d_thist, d_poff, d_p1, geom);
    PANDA_TRY(mark(2));
    hipLaunchKernelGGL(k_bucket_sort, dim3(geom.H, W), dim3(SORT_THREADS), 0, stream, d_p1, d_poff, d_off, d_sorted, geom, NB);
    PANDA_TRY(mark(3));
    PANDA_TRY(hipMemsetAsync(d_bacc, 0, sz_bacc, stream));
    hipLaunchKernelGGL(k_accumulate<Fq>, dim3((chunks + 127) / 128, W), dim3(128), 0, stream, d_bases, d_sorted, d_off, d_bacc, d_parts, log_n, NB, K,
                       chunks);
    PANDA_TRY(mark(4));
    PANDA_TRY(hipMemsetAsync(d_lcount, 0, sz_lcount, stream));
    hipLaunchKernelGGL(k_fixup<Fq>, dim3((NB + 127) / 128, W), dim3(128), 0, stream, d_off, d_parts, d_bacc, NB, K, chunks, d_lcount, d_llist, long_cap);
    hipLaunchKernelGGL(k_fixup_long<Fq>, dim3(LONG_BLOCKS, W), dim3(256), 0, stream, d_parts, d_bacc, NB, chunks, d_lcount, d_llist, long_cap);
    PANDA_TRY(mark(5));
    hipLaunchKernelGGL(k_reduce_groups<Fq>, dim3((groups + 127) / 128, W), dim3(128), 0, stream, d_bacc, d_gsum, NB, groups);
    hipLaunchKernelGGL(k_tree_reduce<Fq>, dim3(lvl1, W), dim3(256), 0, stream, d_gsum, d_l1, groups, per1);
    hipLaunchKernelGGL(k_tree_reduce<Fq>, dim3(1, W), dim3(256), 0, stream, d_l1, d_win, lvl1, lvl1);
    PANDA_TRY(mark(6));
    PANDA_TRY(hipGetLastError());

    std::vector<u32> h_win((size_t)W * PW);
    PANDA_TRY(hipMemcpyAsync(h_win.data(), d_win, (size_t)W * PW * 4, hipMemcpyDeviceToHost, stream));
    PANDA_TRY(hipStreamSynchronize(stream));

    std::vector<Xyzz<Fq>> windows(W);
    for (unsigned w = 0; w < W; w++) {
        const u32 *src = h_win.data() + (size_t)w * PW;
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
    PANDA_TRY(mark(7));
    PANDA_TRY(hipStreamSynchronize(stream));

    float ms = 0;
    for (int i = 0; i < 6; i++) {
        (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
        g_phase_ms[i] = ms;
    }
    (void)hipEventElapsedTime(&ms, ev[6], ev[7]);
    g_phase_ms[6] = ms;
    (void)hipEventElapsedTime(&ms, ev[0], ev[6]);
    g_phase_ms[7] = ms;
    (void)LR;
    return hipSuccess;
}

} // namespace

// ------------------------------------------------------------------------------- C ABI

extern "C" {

panda_error panda_msm_setup_bn254(void) { return panda_success; }
panda_error panda_msm_setup_bls12_377(void) { return panda_success; }

panda_error panda_msm_tear_down(void) { return static_cast<panda_error>(panda::release_thread_arena()); }

panda_error panda_msm_register_bases(unsigned curve, const void *d_bases, unsigned log_n, panda_stream stream)
{
    if (curve > 1 || !d_bases || log_n > 26) return panda_error_invalid_value;
    if (lookup_registered(d_bases, log_n, curve)) return panda_success;
    const u64 n = (u64)1 << log_n;
    const size_t bytes = n * (curve == 0 ? 64 : 96);
    RegisteredBases r{d_bases, nullptr, log_n, curve, -1};
    hipError_t e = hipGetDevice(&r.device);
    if (e == hipSuccess) e = hipMalloc(&r.converted, bytes);
    if (e != hipSuccess) return static_cast<panda_error>(e);
    hipStream_t s = static_cast<hipStream_t>(stream.handle);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (curve == 0)
        hipLaunchKernelGGL(k_convert_bases<Bn254Fq>, dim3(blocks), dim3(256), 0, s, (const u32 *)d_bases, (u32 *)r.converted, n);
    else
        hipLaunchKernelGGL(k_convert_bases<Bls377Fq>, dim3(blocks), dim3(256), 0, s, (const u32 *)d_bases, (u32 *)r.converted, n);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        (void)hipFree(r.converted);
        return static_cast<panda_error>(e);
    }
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    g_registry.push_back(r);
    return panda_success;
}

panda_error panda_msm_unregister_bases(const void *d_bases)
{
    std::vector<void *> to_free;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        for (size_t i = 0; i < g_registry.size();) {
            if (g_registry[i].wire == d_bases) {
                to_free.push_back(g_registry[i].converted);
                g_registry.erase(g_registry.begin() + i);
            } else
                i++;
        }
    }
    if (to_free.empty()) return panda_error_invalid_value;
    (void)hipDeviceSynchronize();
    hipError_t e = hipSuccess;
    for (void *p : to_free) {
        hipError_t f = hipFree(p);
        if (f != hipSuccess) e = f;
    }
    return static_cast<panda_error>(e);
}

panda_error panda_msm_execute_bn254(const panda_msm_configuration cfg) { return static_cast<panda_error>(msm_execute<CurveBn254>(cfg)); }

panda_error panda_msm_execute_bls12_377(const panda_msm_configuration cfg) { return static_cast<panda_error>(msm_execute<CurveBls377>(cfg)); }

panda_error panda_msm_set_window_bits(unsigned window_bits)
{
    if (window_bits > 16) return panda_error_invalid_value;
    g_window_override = window_bits;
    return panda_success;
}

panda_error panda_msm_last_phase_ms(float *ms)
{
    if (!ms) return panda_error_invalid_value;
    for (int i = 0; i < PANDA_MSM_PHASES; i++) ms[i] = g_phase_ms[i];
    return panda_success;
}

const char *panda_msm_phase_name(unsigned phase) { return phase < PANDA_MSM_PHASES ? kPhaseNames[phase] : ""; }

} // extern "C"
