// msm_sort.hip -- scalars -> signed window digits -> per-bucket point lists, for gfx950.
//
// Replaces the reference's three global-atomic passes (calc_lens / allo_arrs / fill_arrs,
// src/cuda/core/unit/msm/msm_cuda.cuh:159-282) and its in-place scalar conversion (:148-157).
// Everything here is an MSD partition sort staged in LDS: per-tile LDS histograms and cursors, tiles grouped in LDS
// and written out as runs, no global atomics anywhere (the first version's global-atomic histogram + scatter cost
// 33 ms at 2^24).
//
//   plain mode   (bases as the caller gave them): per window, level 1 splits the bucket id's high bits into
//                partitions, level 2 gives each partition to a workgroup that ranks the low bits.
//   tabled mode  (precomputed 2^lo[k] * P tables): all windows share one bucket space of up to 2^22 buckets, so the
//                key is three digits deep: level 1 (per window), level 2 (per level-1 partition, ragged tiles),
//                level 3 (one workgroup per (hi, mid) cell merges the W per-window runs and ranks the low bits).
#include <algorithm>
#include <atomic>
#include <cmath>

#include "fe29.h"
#include "msm_sort.h"

using namespace panda29;

namespace {

constexpr unsigned SORT_TILE = 8192; // entries per workgroup tile
constexpr unsigned MAX_PARTS = 1024;
constexpr unsigned SORT_THREADS = 1024; // 16 waves per workgroup: these kernels wait on LDS atomics and HBM, they need the occupancy
constexpr unsigned BS_CHUNK = 8192;

template <class Code>
struct CodeTraits;
template <>
struct CodeTraits<uint16_t> {
    static constexpr u32 ZERO = 0x7fffu; // "+2^15" cannot occur with the recoding below, so it encodes digit 0
    static constexpr u32 MAG = 0x7fffu;
    static constexpr int SIGN = 15;
};
template <>
struct CodeTraits<u32> {
    static constexpr u32 ZERO = 0x7fffffffu;
    static constexpr u32 MAG = 0x7fffffffu;
    static constexpr int SIGN = 31;
};

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

// workgroup 0 of the digits kernel (256 threads, four per sampled row); every thread of the workgroup must call it
__device__ __forceinline__ void check_samples(const panda::SampleCheck &sc)
{
    int differs = 0;
    for (unsigned t = threadIdx.x >> 2; t < panda::REG_SAMPLES; t += blockDim.x >> 2) {
        const u64 row = panda::sample_row(t, sc.n);
        for (unsigned k = threadIdx.x & 3u; k < sc.row_words; k += 4) differs |= sc.samples[t * sc.row_words + k] != sc.wire[row * sc.row_words + k];
    }
    differs = __syncthreads_or(differs);
    if (threadIdx.x == 0) {
        *sc.flag_dev = differs ? 1u : 0u;
        if (differs) *sc.flag_host = 1u;
    }
}

// scalar (Montgomery wire form) -> W signed digits.  code = (neg << SIGN) | (|d| - 1), ZERO for d = 0.
// Replaces init_handle_scalars_kernel + the slice extraction of calc_lens/fill_arrs (msm_cuda.cuh:148-205,232-282);
// the scalars are only read.
template <class Fr, class Code>
__global__ void __launch_bounds__(256) k_digits(const u32 *__restrict__ scalars, Code *__restrict__ dig, u64 n, panda::WindowPlan plan, panda::SampleCheck sc)
{
    typedef CodeTraits<Code> CT;
    constexpr int L = Fr::L;
    if (blockIdx.x == 0 && sc.wire) check_samples(sc);
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
            code = mag ? ((1u << CT::SIGN) | (mag - 1)) : CT::ZERO;
        } else {
            carry = 0;
            code = raw ? (raw - 1) : CT::ZERO;
        }
        dig[(u64)k * n + i] = (Code)code;
    }
}

// k_digits and the level-1 histogram (k_part_hist) in one pass over the scalars: a block takes one tile of SORT_TILE
// scalars and counts every window's partitions in LDS (W * H counters, dynamic shared memory), so the digit codes are
// not read back for counting.  Used for large inputs, where the tiles alone fill the chip.
struct DigitsHistGeom {
    unsigned lo_bits, H, tiles;
};

template <class Fr, class Code>
__global__ void __launch_bounds__(1024) k_digits_hist(const u32 *__restrict__ scalars, Code *__restrict__ dig, u32 *__restrict__ tile_hist, u64 n,
                                                     panda::WindowPlan plan, DigitsHistGeom g, panda::SampleCheck sc)
{
    typedef CodeTraits<Code> CT;
    constexpr int L = Fr::L;
    extern __shared__ u32 hist[]; // [W][H]
    if (blockIdx.x == 0 && sc.wire) check_samples(sc);
    const unsigned tile = blockIdx.x, tid = threadIdx.x;
    const unsigned WH = plan.W * g.H, threads = blockDim.x;
    for (unsigned i = tid; i < WH; i += threads) hist[i] = 0;
    __syncthreads();
    const u64 begin = (u64)tile * SORT_TILE, end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
    for (u64 i = begin + tid; i < end; i += threads) {
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
            if (raw >= half) {
                u32 mag = full - raw;
                carry = 1;
                code = mag ? ((1u << CT::SIGN) | (mag - 1)) : CT::ZERO;
            } else {
                carry = 0;
                code = raw ? (raw - 1) : CT::ZERO;
            }
            dig[(u64)k * n + i] = (Code)code;
            if (code != CT::ZERO) atomicAdd(&hist[k * g.H + ((code & CT::MAG) >> g.lo_bits)], 1u);
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < WH; i += threads) {
        const unsigned k = i / g.H, h = i - k * g.H;
        tile_hist[((u64)k * g.tiles + tile) * g.H + h] = hist[i];
    }
}

// codes of window `dw` at positions [begin, end), 16 bytes per load (8 u16 or 4 u32 codes); begin is a multiple of the
// tile size, the arrays are padded, codes past `end` are skipped by index.  f(index, code) for every code in range.
template <class Code, class Fn>
__device__ __forceinline__ void for_each_code(const Code *__restrict__ dw, u64 begin, u64 end, Fn f)
{
    constexpr unsigned PER = 16 / sizeof(Code);
    if (reinterpret_cast<uintptr_t>(dw + begin) & 15) { // rows of fewer than PER codes (n < 8 or 4) need not start on 16 bytes: one code per load
        for (u64 a = begin + threadIdx.x; a < end; a += blockDim.x) f(a, (u32)dw[a]);
        return;
    }
    for (u64 a = begin + (u64)threadIdx.x * PER; a < end; a += (u64)blockDim.x * PER) {
        const uint4 v = *reinterpret_cast<const uint4 *>(dw + a);
        const u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (unsigned j = 0; j < PER; j++) {
            const u32 code = sizeof(Code) == 2 ? (wv[j >> 1] >> (16 * (j & 1))) & 0xffffu : wv[j];
            if (a + j < end) f(a + j, code);
        }
    }
}

// ---- level 1 (both modes): per window, bucket id high bits -> partition ------------------------------------------
// Bucket ids are split into `hi` (partition) and `lo` bits.  Level 1 moves every [lo][sign][point id] word into its
// partition with per-tile LDS histograms and LDS cursors.  A partition's output range equals its input range, so no
// global prefix over the buckets is needed.

struct SortGeom {
    unsigned log_n, lo_bits, H, tiles;
    u32 row0; // plain mode: added to every point id in the final entries (the chunk's first row of the base array)
};

template <class Code>
__global__ void __launch_bounds__(256) k_part_hist(const Code *__restrict__ dig, u32 *__restrict__ tile_hist, SortGeom g)
{
    typedef CodeTraits<Code> CT;
    __shared__ u32 h[MAX_PARTS];
    const unsigned w = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    for (unsigned i = tid; i < g.H; i += 256) h[i] = 0;
    __syncthreads();
    const u64 n = (u64)1 << g.log_n;
    const Code *dw = dig + ((u64)w << g.log_n);
    const u64 begin = (u64)tile * SORT_TILE, end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
    for_each_code<Code>(dw, begin, end, [&](u64, u32 code) {
        if (code != CT::ZERO) atomicAdd(&h[(code & CT::MAG) >> g.lo_bits], 1u);
    });
    __syncthreads();
    u32 *out = tile_hist + ((u64)w * g.tiles + tile) * g.H;
    for (unsigned i = tid; i < g.H; i += 256) out[i] = h[i];
}

// tile_hist -> exclusive prefix over tiles (tile_pref), one WAVE per (window, partition) column: 64 tiles per step with a
// shuffle scan instead of one dependent load per tile; column totals go to `totals`.  tile_hist keeps the counts: the
// scatter kernels start from them instead of counting their tile again.
__global__ void __launch_bounds__(1024) k_part_scan_cols(const u32 *__restrict__ tile_hist, u32 *__restrict__ tile_pref, u32 *__restrict__ totals, SortGeom g)
{
    const unsigned w = blockIdx.y, lane = threadIdx.x & 63, h = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (h >= g.H) return; // whole wave exits together
    const u64 col = (u64)w * g.tiles * g.H + h;
    u32 run = 0;
    for (unsigned t0 = 0; t0 < g.tiles; t0 += 64) {
        const unsigned tile = t0 + lane;
        u32 v = tile < g.tiles ? tile_hist[col + (u64)tile * g.H] : 0;
        u32 inc = v;
        for (unsigned d = 1; d < 64; d <<= 1) {
            u32 up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        if (tile < g.tiles) tile_pref[col + (u64)tile * g.H] = run + inc - v;
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

// block-wide exclusive scan of `count` LDS words in place by THREADS threads
template <unsigned THREADS = SORT_THREADS>
__device__ __forceinline__ void block_exclusive_scan(u32 *a, unsigned count, u32 *scratch /* THREADS words */)
{
    const unsigned tid = threadIdx.x;
    const unsigned per = (count + THREADS - 1) / THREADS;
    u32 local = 0;
    for (unsigned j = 0; j < per; j++) {
        unsigned i = tid * per + j;
        if (i < count) local += a[i];
    }
    scratch[tid] = local;
    __syncthreads();
    for (unsigned d = 1; d < THREADS; d <<= 1) {
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
// stores runs of consecutive addresses instead of 64 unrelated words.
template <class Code, class Word>
__global__ void __launch_bounds__(SORT_THREADS) k_part_scatter(const Code *__restrict__ dig, const u32 *__restrict__ tile_hist, const u32 *__restrict__ tile_pref,
                                                      const u32 *__restrict__ part_off, Word *__restrict__ p1, SortGeom g)
{
    typedef CodeTraits<Code> CT;
    __shared__ u32 lstart[MAX_PARTS]; // local start of each partition's run in the staging buffer
    __shared__ u32 lcur[MAX_PARTS];   // local cursor
    __shared__ u32 gbase[MAX_PARTS];  // global position of this tile's first element of the partition
    __shared__ Word words[SORT_TILE];
    __shared__ uint16_t parts_of[SORT_TILE];
    __shared__ u32 scratch[SORT_THREADS];
    const unsigned w = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const u64 row = ((u64)w * g.tiles + tile) * g.H;
    const u32 *po = part_off + (u64)w * (g.H + 1);
    for (unsigned i = tid; i < g.H; i += SORT_THREADS) {
        lstart[i] = tile_hist[row + i]; // this tile's count, turned into a local offset by the scan below
        lcur[i] = 0;
        gbase[i] = po[i] + tile_pref[row + i];
    }
    __syncthreads();
    const u64 n = (u64)1 << g.log_n;
    const Code *dw = dig + ((u64)w << g.log_n);
    Word *pw = p1 + ((u64)w << g.log_n);
    const u64 begin = (u64)tile * SORT_TILE, end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
    const u32 lo_mask = (1u << g.lo_bits) - 1;
    block_exclusive_scan(lstart, g.H, scratch);
    for_each_code<Code>(dw, begin, end, [&](u64 i, u32 code) {
        if (code == CT::ZERO) return;
        u32 b = code & CT::MAG, h = b >> g.lo_bits;
        u32 slot = lstart[h] + atomicAdd(&lcur[h], 1u);
        words[slot] = ((Word)(b & lo_mask) << (g.log_n + 1)) | (Word)(((code >> CT::SIGN) << g.log_n) | (u32)i);
        parts_of[slot] = (uint16_t)h;
    });
    __syncthreads();
    const u32 total = lstart[g.H - 1] + lcur[g.H - 1];
    for (u32 j = tid; j < total; j += SORT_THREADS) {
        u32 h = parts_of[j];
        pw[gbase[h] + (j - lstart[h])] = words[j];
    }
}

// Tabled mode, level 1: the same scatter, but the word is stored split -- a u32 [b3 key bits][sign][point id], which is
// exactly the word level 2 hands on, and a u8 with the b2 key bits level 2 partitions by.  Level 2 then counts from the
// byte array alone and never moves more than five bytes per entry.
template <class Code>
__global__ void __launch_bounds__(SORT_THREADS) k1_scatter_split(const Code *__restrict__ dig, const u32 *__restrict__ tile_hist, const u32 *__restrict__ tile_pref,
                                                        const u32 *__restrict__ part_off, u32 *__restrict__ p1_lo, unsigned char *__restrict__ p1_hi, SortGeom g,
                                                        unsigned b3)
{
    typedef CodeTraits<Code> CT;
    __shared__ u32 lstart[MAX_PARTS];
    __shared__ u32 lcur[MAX_PARTS];
    __shared__ u32 gbase[MAX_PARTS];
    __shared__ u32 words[SORT_TILE];
    __shared__ unsigned char his[SORT_TILE];
    __shared__ uint16_t parts_of[SORT_TILE];
    __shared__ u32 scratch[SORT_THREADS];
    const unsigned w = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const u64 row = ((u64)w * g.tiles + tile) * g.H;
    const u32 *po = part_off + (u64)w * (g.H + 1);
    for (unsigned i = tid; i < g.H; i += SORT_THREADS) {
        lstart[i] = tile_hist[row + i];
        lcur[i] = 0;
        gbase[i] = po[i] + tile_pref[row + i];
    }
    __syncthreads();
    const u64 n = (u64)1 << g.log_n;
    const Code *dw = dig + ((u64)w << g.log_n);
    u32 *plo = p1_lo + ((u64)w << g.log_n);
    unsigned char *phi = p1_hi + ((u64)w << g.log_n);
    const u64 begin = (u64)tile * SORT_TILE, end = begin + SORT_TILE < n ? begin + SORT_TILE : n;
    const u32 b3_mask = (1u << b3) - 1, b2_mask = (1u << (g.lo_bits - b3)) - 1;
    block_exclusive_scan(lstart, g.H, scratch);
    for_each_code<Code>(dw, begin, end, [&](u64 i, u32 code) {
        if (code == CT::ZERO) return;
        u32 b = code & CT::MAG, h = b >> g.lo_bits;
        u32 slot = lstart[h] + atomicAdd(&lcur[h], 1u);
        words[slot] = ((b & b3_mask) << (g.log_n + 1)) | ((code >> CT::SIGN) << g.log_n) | (u32)i;
        his[slot] = (unsigned char)((b >> b3) & b2_mask);
        parts_of[slot] = (uint16_t)h;
    });
    __syncthreads();
    const u32 total = lstart[g.H - 1] + lcur[g.H - 1];
    for (u32 j = tid; j < total; j += SORT_THREADS) {
        const u32 h = parts_of[j];
        const u32 dst = gbase[h] + (j - lstart[h]);
        plo[dst] = words[j];
        phi[dst] = his[j];
    }
}

// inclusive scan of cnt[0..127] in place by the first two waves of the block: shuffles inside a wave, one barrier
// to pass wave 0's total on (instead of seven barrier-separated Hillis-Steele steps).  All threads must call it.
__device__ __forceinline__ void scan128_inclusive(u32 *cnt, u32 *carry /* one LDS word */)
{
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    u32 v = tid < 128 ? cnt[tid] : 0;
    if (tid < 128) {
        for (unsigned d = 1; d < 64; d <<= 1) {
            const u32 up = __shfl_up(v, d, 64);
            if (lane >= d) v += up;
        }
        if (tid == 63) *carry = v;
    }
    __syncthreads();
    if (tid >= 64 && tid < 128) v += *carry;
    if (tid < 128) cnt[tid] = v;
    __syncthreads();
}

// ---- plain mode, level 2: one workgroup per (partition, window) counts the lo values, publishes the bucket offsets,
// then ranks the ids chunk by chunk in LDS and writes each chunk out as runs
__global__ void __launch_bounds__(SORT_THREADS) k_bucket_sort(const u32 *__restrict__ p1, const u32 *__restrict__ part_off, u32 *__restrict__ off,
                                                     u32 *__restrict__ sorted, SortGeom g, unsigned NB)
{
    __shared__ u32 cnt[128], cur[128], lstart[128], lcur[128], scan_carry;
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
    scan128_inclusive(cnt, &scan_carry);
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
        scan128_inclusive(lstart, &scan_carry);
        if (tid < 128) {
            lstart[tid] -= c0; // exclusive
            cnt[tid] = c0;
        }
        __syncthreads();
        for (u32 j = cbeg + tid; j < cend; j += SORT_THREADS) {
            u32 v = pw[j];
            u32 l = v >> shift;
            u32 slot = lstart[l] + atomicAdd(&lcur[l], 1u);
            words[slot] = ((v & id_mask) + g.row0) | (((v >> g.log_n) & 1u) << 31);
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

// ---- tabled mode, level 2: ragged segments ---------------------------------------------------------------------------
// Segment s = h1 * W + k is level-1 partition h1 of window k: positions [k*n + part_off[k][h1], k*n + part_off[k][h1+1])
// of p1.  Its tiles are numbered seg_tile[s] .. seg_tile[s+1]-1; the launch covers an upper bound of tiles and a
// workgroup finds its segment by binary search.  Segments are numbered partition-major so that a RANGE of level-1 partitions
// -- a contiguous range of buckets -- is a contiguous range of segments, of tiles and of level-3 cells: levels 2 and 3 can then
// run for the front of the bucket space and for the rest in separate launches (SortSplit, msm_sort.h).

struct TabledGeom {
    unsigned log_n, W;
    unsigned b1, b2, b3; // bucket id = [b1][b2][b3]
    unsigned H1, H2, S;  // 2^b1, 2^b2, W * H1
    unsigned max_tiles2; // launch bound for the level-2 tile kernels
    unsigned Q;          // H1 * H2 cells
    unsigned row_shift;  // final entries name row (k << row_shift) + row0 + i of the concatenated tables: log2 of the rows per table
    u32 row0;            // and the first row of the point range being sorted (both = log_n, 0 unless a call runs in point-range chunks)
    unsigned per_window; // 0: one list, level 3 merges the W windows of a cell (tabled mode);  1: W lists, a level-3 cell is one window's
                         // (plain mode with windows too wide for the two-level sort): cells are numbered window-major, entries name row0 + i
};
__host__ __device__ __forceinline__ unsigned seg_of(const TabledGeom &g, unsigned k, unsigned h1) { return h1 * g.W + k; }

// what one launch of the level-2 / level-3 kernels covers: segments [s_lo, s_hi), cells [q_lo, q_hi) (q_lo a multiple of 1024)
struct SortRange {
    unsigned s_lo, s_hi, q_lo, q_hi;
};

// one workgroup: seg_tile[0..S] = exclusive prefix of ceil(len_s / SORT_TILE)
__global__ void __launch_bounds__(SORT_THREADS) k2_seg_tiles(const u32 *__restrict__ part_off, u32 *__restrict__ seg_tile, TabledGeom g)
{
    __shared__ u32 scratch[SORT_THREADS];
    const unsigned tid = threadIdx.x;
    const unsigned per = (g.S + SORT_THREADS - 1) / SORT_THREADS;
    u32 local = 0;
    for (unsigned j = 0; j < per; j++) {
        unsigned s = tid * per + j;
        if (s < g.S) {
            unsigned k = s % g.W, h1 = s / g.W;
            const u32 *po = part_off + (u64)k * (g.H1 + 1);
            local += (po[h1 + 1] - po[h1] + SORT_TILE - 1) / SORT_TILE;
        }
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
        unsigned s = tid * per + j;
        if (s < g.S) {
            unsigned k = s % g.W, h1 = s / g.W;
            const u32 *po = part_off + (u64)k * (g.H1 + 1);
            seg_tile[s] = run;
            run += (po[h1 + 1] - po[h1] + SORT_TILE - 1) / SORT_TILE;
        }
    }
    if (tid == SORT_THREADS - 1) seg_tile[g.S] = scratch[SORT_THREADS - 1];
}

struct TileRange {
    unsigned s;     // segment
    u64 begin, end; // positions in p1 / p2
    u64 seg_begin;
};

// workgroup `rel` of a launch over the segments [R.s_lo, R.s_hi) -> (tile, segment, range); false if the launch has more workgroups
// than the range has tiles (uniform over the workgroup)
__device__ __forceinline__ bool locate_tile(TileRange &r, unsigned &tile, unsigned rel, const u32 *seg_tile, const u32 *part_off, const TabledGeom &g, const SortRange &R)
{
    tile = seg_tile[R.s_lo] + rel;
    if (tile >= seg_tile[R.s_hi]) return false;
    unsigned lo = R.s_lo, hi = R.s_hi; // invariant: seg_tile[lo] <= tile < seg_tile[hi]
    while (hi - lo > 1) {
        unsigned mid = (lo + hi) >> 1;
        if (seg_tile[mid] <= tile) lo = mid;
        else hi = mid;
    }
    r.s = lo;
    const unsigned k = lo % g.W, h1 = lo / g.W;
    const u32 *po = part_off + (u64)k * (g.H1 + 1);
    r.seg_begin = ((u64)k << g.log_n) + po[h1];
    const u64 seg_end = ((u64)k << g.log_n) + po[h1 + 1];
    r.begin = r.seg_begin + (u64)(tile - seg_tile[lo]) * SORT_TILE;
    r.end = r.begin + SORT_TILE < seg_end ? r.begin + SORT_TILE : seg_end;
    return true;
}

__global__ void __launch_bounds__(256) k2_hist(const unsigned char *__restrict__ p1_hi, const u32 *__restrict__ part_off, const u32 *__restrict__ seg_tile,
                                               u32 *__restrict__ tile_hist, TabledGeom g, SortRange R)
{
    __builtin_amdgcn_s_setprio(3); // beside the bucket accumulation of a split sort (SortSplit) these waves must win the issue arbitration: they issue little, it issues always
    __shared__ u32 h[256];
    const unsigned tid = threadIdx.x;
    unsigned tile;
    TileRange r;
    if (!locate_tile(r, tile, blockIdx.x, seg_tile, part_off, g, R)) return;
    h[tid] = 0;
    __syncthreads();
    // 16 entries per load: aligned 16-byte vectors from the vector that contains `begin` on; bytes outside the tile are
    // skipped by index (the array is padded by 16 bytes, so the last vector may run past `end`)
    const u64 first = r.begin & ~(u64)15;
    for (u64 a = first + (u64)tid * 16; a < r.end; a += 256 * 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p1_hi + a);
        const u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const u64 i = a + j;
            if (i >= r.begin && i < r.end) atomicAdd(&h[(wv[j >> 2] >> (8 * (j & 3))) & 0xffu], 1u);
        }
    }
    __syncthreads();
    if (tid < g.H2) tile_hist[(u64)tile * g.H2 + tid] = h[tid];
}

// one WAVE per (segment, h2) column: exclusive prefix over the segment's tiles (tile_pref), column total to totals[s][h2]
__global__ void __launch_bounds__(256) k2_scan_cols(const u32 *__restrict__ tile_hist, u32 *__restrict__ tile_pref, const u32 *__restrict__ seg_tile,
                                                    u32 *__restrict__ totals, TabledGeom g, SortRange R)
{
    __builtin_amdgcn_s_setprio(3); // beside the bucket accumulation of a split sort (SortSplit) these waves must win the issue arbitration: they issue little, it issues always
    // four waves, four columns (a wave per SIMD: sixteen-wave workgroups of this kernel took 3.9 ms beside the accumulation, 0.03 alone)
    const unsigned s = R.s_lo + blockIdx.y, lane = threadIdx.x & 63, h = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= g.H2) return; // whole wave exits together
    const unsigned t_begin = seg_tile[s], t_end = seg_tile[s + 1];
    u32 run = 0;
    for (unsigned t0 = t_begin; t0 < t_end; t0 += 64) {
        const unsigned tile = t0 + lane;
        u32 v = tile < t_end ? tile_hist[(u64)tile * g.H2 + h] : 0;
        u32 inc = v;
        for (unsigned d = 1; d < 64; d <<= 1) {
            u32 up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        if (tile < t_end) tile_pref[(u64)tile * g.H2 + h] = run + inc - v;
        run += __shfl(inc, 63, 64);
    }
    if (lane == 0) totals[(u64)s * g.H2 + h] = run;
}

// one block per segment: sub_off[s][0..H2] = exclusive scan of the column totals (relative to the segment start)
__global__ void __launch_bounds__(256) k2_offsets(const u32 *__restrict__ totals, u32 *__restrict__ sub_off, TabledGeom g, SortRange R)
{
    __builtin_amdgcn_s_setprio(3); // beside the bucket accumulation of a split sort (SortSplit) these waves must win the issue arbitration: they issue little, it issues always
    __shared__ u32 tot[256];
    const unsigned s = R.s_lo + blockIdx.x, t = threadIdx.x;
    const u32 mine = t < g.H2 ? totals[(u64)s * g.H2 + t] : 0;
    tot[t] = mine;
    __syncthreads();
    for (unsigned d = 1; d < 256; d <<= 1) {
        u32 v = (t >= d) ? tot[t - d] : 0;
        __syncthreads();
        tot[t] += v;
        __syncthreads();
    }
    if (t < g.H2) sub_off[(u64)s * (g.H2 + 1) + t] = tot[t] - mine;
    if (t == 255) sub_off[(u64)s * (g.H2 + 1) + g.H2] = tot[255];
}

// level-2 scatter: the u32 halves of the level-1 words, grouped by their u8 halves within the segment
// THREADS = 256 (a wave per SIMD): the variant that finds room BESIDE the bucket accumulation when it runs for the rest of a split sort
// (sixteen-wave workgroups are only placed when accumulate workgroups retire: profiles/r05_overlap_sort_accumulate.txt)
template <unsigned THREADS>
__global__ void __launch_bounds__(THREADS) k2_scatter(const u32 *__restrict__ p1_lo, const unsigned char *__restrict__ p1_hi, const u32 *__restrict__ part_off,
                                                  const u32 *__restrict__ seg_tile, const u32 *__restrict__ tile_hist, const u32 *__restrict__ tile_pref,
                                                  const u32 *__restrict__ sub_off, u32 *__restrict__ p2, TabledGeom g, SortRange R)
{
    __builtin_amdgcn_s_setprio(3); // beside the bucket accumulation of a split sort (SortSplit) these waves must win the issue arbitration: they issue little, it issues always
    __shared__ u32 lstart[256], lcur[256];
    __shared__ u64 gbase[256];
    __shared__ u32 words[SORT_TILE];
    __shared__ unsigned char parts_of[SORT_TILE];
    __shared__ u32 scratch[THREADS];
    const unsigned tid = threadIdx.x;
    unsigned tile;
    TileRange r;
    if (!locate_tile(r, tile, blockIdx.x, seg_tile, part_off, g, R)) return;
    if (tid < 256) {
        lstart[tid] = tid < g.H2 ? tile_hist[(u64)tile * g.H2 + tid] : 0;
        lcur[tid] = 0;
        if (tid < g.H2) gbase[tid] = r.seg_begin + sub_off[(u64)r.s * (g.H2 + 1) + tid] + tile_pref[(u64)tile * g.H2 + tid];
    }
    __syncthreads();
    block_exclusive_scan<THREADS>(lstart, g.H2, scratch);
    // 16 entries per thread and step: one aligned 16-byte vector of keys and the four 16-byte vectors of words that go with
    // it (both arrays are padded, entries outside the tile are skipped by index)
    const u64 first = r.begin & ~(u64)15;
    for (u64 a = first + (u64)tid * 16; a < r.end; a += (u64)THREADS * 16) {
        const uint4 kv = *reinterpret_cast<const uint4 *>(p1_hi + a);
        const u32 keys[4] = {kv.x, kv.y, kv.z, kv.w};
        const uint4 *lo4 = reinterpret_cast<const uint4 *>(p1_lo + a);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 wv = lo4[q];
            const u32 ws[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u64 i = a + 4 * q + j;
                if (i >= r.begin && i < r.end) {
                    const u32 h = (keys[q] >> (8 * j)) & 0xffu;
                    const u32 slot = lstart[h] + atomicAdd(&lcur[h], 1u);
                    words[slot] = ws[j];
                    parts_of[slot] = (unsigned char)h;
                }
            }
        }
    }
    __syncthreads();
    const u32 total = (u32)(r.end - r.begin);
    for (u32 j = tid; j < total; j += THREADS) {
        u32 h = parts_of[j];
        p2[gbase[h] + (j - lstart[h])] = words[j];
    }
}

// ---- tabled mode, level 3: merge the windows -------------------------------------------------------------------------
// Cell q = h1 * H2 + h2 owns buckets [q << b3, (q+1) << b3).  Its entries are the W runs (k, h1, h2) of p2.

__device__ __forceinline__ void cell_run(u64 &begin, u32 &len, unsigned k, unsigned h1, unsigned h2, const u32 *part_off, const u32 *sub_off, const TabledGeom &g)
{
    const unsigned s = seg_of(g, k, h1);
    const u32 *so = sub_off + (u64)s * (g.H2 + 1);
    begin = ((u64)k << g.log_n) + part_off[(u64)k * (g.H1 + 1) + h1] + so[h2];
    len = so[h2 + 1] - so[h2];
}

// cell_cnt[q] = entries of cell q; blk_sum[b] = sum over the 1024 cells of block b.  per_window: cell q = k * Q + (h1 * H2 + h2)
__device__ __forceinline__ unsigned cells_total(const TabledGeom &g) { return g.per_window ? g.W * g.Q : g.Q; }

// A block of 1024 cells per workgroup of 1024 threads -- or of 256 (one wave per SIMD) when the kernel runs for the rest of a split sort
// and has to find room beside the bucket accumulation (see k3_merge_small)
__global__ void __launch_bounds__(1024) k3_cell_counts(const u32 *__restrict__ sub_off, u32 *__restrict__ cell_cnt, u32 *__restrict__ blk_sum, TabledGeom g, SortRange R)
{
    __builtin_amdgcn_s_setprio(3); // beside the bucket accumulation of a split sort (SortSplit) these waves must win the issue arbitration: they issue little, it issues always
    __shared__ u32 red[16];
    const unsigned blk = R.q_lo / 1024 + blockIdx.x, t = threadIdx.x;
    const unsigned qbits = g.b1 + g.b2; // H2 and Q = H1 H2 are powers of two
    u32 sum = 0;
#pragma unroll 1
    for (unsigned q = blk * 1024 + t; q < blk * 1024 + 1024 && q < R.q_hi; q += blockDim.x) {
        const unsigned cell = g.per_window ? q & ((1u << qbits) - 1) : q;
        const unsigned h1 = cell >> g.b2, h2 = cell & (g.H2 - 1);
        const unsigned k_begin = g.per_window ? q >> qbits : 0u, k_end = g.per_window ? k_begin + 1 : g.W;
        u32 total = 0;
#pragma unroll 1
        for (unsigned k = k_begin; k < k_end; k++) {
            const unsigned at = seg_of(g, k, h1) * (g.H2 + 1) + h2; // < 2^14 segments x 257
            total += sub_off[at + 1] - sub_off[at];
        }
        cell_cnt[q] = total;
        sum += total;
    }
    for (unsigned d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
    if ((t & 63) == 0) red[t >> 6] = sum;
    __syncthreads();
    if (t == 0) {
        u32 v = 0;
        for (unsigned i = 0; i < blockDim.x / 64; i++) v += red[i];
        blk_sum[blk] = v;
    }
}

// cell_off[q] = entries of all earlier cells: earlier blocks (blk_sum) + exclusive scan inside the block of 1024 cells.  blockDim.x = 1024
// (a cell per thread) or 256 (four consecutive cells per thread: the launch that has to find room beside the bucket accumulation).
__global__ void __launch_bounds__(1024) k3_cell_offsets(const u32 *__restrict__ cell_cnt, const u32 *__restrict__ blk_sum, u32 *__restrict__ cell_off, TabledGeom g, SortRange R)
{
    __builtin_amdgcn_s_setprio(3); // beside the bucket accumulation of a split sort (SortSplit) these waves must win the issue arbitration: they issue little, it issues always
    __shared__ u32 red[1024];
    __shared__ u32 sc[1024];
    const unsigned blk = R.q_lo / 1024 + blockIdx.x, t = threadIdx.x, T = blockDim.x, per = 1024 / T;
    u32 v = 0;
    for (unsigned b = t; b < blk; b += T) v += blk_sum[b]; // the blocks of earlier launches included: their counts are complete (stream / event order)
    red[t] = v;
    const unsigned q0 = blk * 1024 + t * per;
    u32 local = 0;
    for (unsigned j = 0; j < per; j++) local += q0 + j < R.q_hi ? cell_cnt[q0 + j] : 0;
    sc[t] = local;
    __syncthreads();
    for (unsigned s = T >> 1; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    for (unsigned d = 1; d < T; d <<= 1) {
        u32 u = (t >= d) ? sc[t - d] : 0;
        __syncthreads();
        sc[t] += u;
        __syncthreads();
    }
    u32 run = red[0] + sc[t] - local;
    for (unsigned j = 0; j < per; j++) {
        const unsigned q = q0 + j;
        if (q < R.q_hi) {
            cell_off[q] = run;
            run += cell_cnt[q];
            if (q == R.q_hi - 1) cell_off[R.q_hi] = run; // the start of the next launch's first cell (written again, to the same value, by that launch); the list's total at the end
        }
    }
}

constexpr unsigned K3_THREADS = 1024; // (512 threads on half-size cells, four workgroups per CU: k3_merge 0.994 -> 0.968 ms at 2^24, k1_scatter_split 0.535 -> 0.590 with its 256 partitions: round 5)
constexpr unsigned K3_PER = 16;                     // words a thread keeps in registers
constexpr unsigned K3_CAP = K3_THREADS * K3_PER;    // a cell of up to this many entries is read from HBM once; also the entries of the LDS image
constexpr unsigned K3_PER_WIDE = 32;                // the variant for the dense half of the cells where the geometry cannot make them smaller

// measurement builds only (-DPANDA_K3_STAMPS, tools/k3_stamps.sh): s_memrealtime (100 MHz) at the phase boundaries of thread 0 of every 16th
// workgroup of k3_merge, into a buffer nothing else reads -- how the overflowing lower cells were found (profiles/r05_k3_merge_cells.txt)
#ifdef PANDA_K3_STAMPS
__device__ unsigned long long g_k3_stamps[1024 * 8];
#define K3_STAMP(n)                                                                                                          \
    do {                                                                                                                     \
        if (threadIdx.x == 0 && (cell & 15u) == 0 && (cell >> 4) < 1024) g_k3_stamps[(cell >> 4) * 8 + (n)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define K3_STAMP(n)
#endif

// THREADS = 256 (with PER = K3_PER: cells of up to 4096 entries): the sparse half of the cells -- 3 k entries each where the dense half
// holds 21.5 k -- gave a 1024-thread workgroup three entries per thread; four-wave workgroups, eight to a CU, merge them in a fifth of the time.
// PER = K3_PER_WIDE: a cell of up to 2 K3_CAP entries is still read once -- 32 words and their ranks in registers (one workgroup per CU
// instead of two) -- and leaves through the same LDS image in two rounds.  For the cells of the lower half of the bucket space at 2^25 and
// 2^26 points, where the entry word leaves b3 only 5 or 6 bits and a smaller cell would make level 1 a 512-way partition.
template <unsigned PER, unsigned THREADS = K3_THREADS>
__global__ void __launch_bounds__(THREADS) k3_merge(const u32 *__restrict__ p2, const u32 *__restrict__ part_off, const u32 *__restrict__ sub_off,
                                              const u32 *__restrict__ cell_off, u32 *__restrict__ off, u32 *__restrict__ sorted, TabledGeom g, unsigned NB, SortRange R)
{
    __shared__ u32 cnt[128], cur[128], scan_carry;
    constexpr unsigned IMG = THREADS * K3_PER; // entries of the LDS image
    __shared__ u32 outbuf[IMG];
    __shared__ u64 rbegin[64];
    __shared__ u32 rlen[64];
    __shared__ u32 vstart[65];
    const unsigned tid = threadIdx.x;
    // tabled mode: cell q of the one list, W runs to merge;  per-window mode: cell q of window k0's own list, one run
    const unsigned cell = R.q_lo + blockIdx.x;
    const unsigned k0 = g.per_window ? cell / g.Q : 0u, q = g.per_window ? cell % g.Q : cell;
    const unsigned nruns = g.per_window ? 1u : g.W;
    const unsigned h1 = q / g.H2, h2 = q % g.H2;
    const unsigned L = 1u << g.b3;
    K3_STAMP(0);
    // position of the cell's first entry within its list, and where the list starts in `sorted` / `off`
    const u32 out_rel = cell_off[cell] - (g.per_window ? cell_off[k0 * g.Q] : 0u);
    // the last cell of a launch that stops short of the list's end also publishes where the NEXT cell's first bucket starts: the
    // accumulation of this launch's part reads off[b + 1] of its last bucket before the launch that owns that cell has run
    const bool closes_part = !g.per_window && cell + 1 == R.q_hi && q + 1 < g.Q;
    u32 *sw = sorted + ((u64)k0 << g.log_n);
    u32 *ow = off + (u64)k0 * (NB + 1);
    if (tid < nruns) {
        u64 b;
        u32 len;
        cell_run(b, len, k0 + tid, h1, h2, part_off, sub_off, g);
        rbegin[tid] = b;
        rlen[tid] = len;
    }
    if (tid < 128) cnt[tid] = 0;
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (unsigned k = 0; k < nruns; k++) {
            vstart[k] = run;
            run += rlen[k];
        }
        vstart[nruns] = run;
    }
    __syncthreads();
    K3_STAMP(1); // setup: the runs of the cell, their prefix, two barriers
    const u32 N = vstart[nruns]; // uniform over the block
    const unsigned shift = g.log_n + 1;
    const u32 id_mask = (1u << g.log_n) - 1;
    // virtual position in the concatenation of the runs -> run
    auto window_of = [&](u32 p) -> unsigned { // branch-free: the broadcast reads pipeline
        unsigned k = 0;
        for (unsigned j = 1; j < nruns; j++) k += p >= vstart[j] ? 1u : 0u;
        return k;
    };
    // final word: row k*n + i of the tables (tabled mode) or row i of the bases (per-window mode), bit 31 = negate
    const unsigned table_shift = g.per_window ? 32u : g.row_shift;
    auto final_word = [&](u32 v, unsigned k) -> u32 {
        return ((v & id_mask) + g.row0 + (table_shift < 32 ? (k << table_shift) : 0u)) | (((v >> g.log_n) & 1u) << 31);
    };

    if (closes_part && tid == 0) ow[(u64)(q + 1) << g.b3] = out_rel + N;
    if (N == 0) { // an empty cell: its buckets all start where the cell does
        if (tid < L) ow[((u64)q << g.b3) + tid] = out_rel;
        if (q == g.Q - 1 && tid == 0) ow[NB] = out_rel;
        return;
    }
    if (N <= THREADS * PER) {
        // one read: words stay in registers, are ranked into LDS in bucket order and leave as one linear, coalesced copy
        // (the cell's buckets are adjacent in the output)
        // ONE LDS atomic per entry: the count's return value is the entry's rank within its bucket, kept (with the bucket's 7 bits) in a
        // register until the offsets are known -- the ranking pass then needs no second atomic.  With 128 counters under 64 lanes the
        // atomics are the merge's bottleneck (two per entry: 1.00 ms at 2^24; one: see profiles/r04_msm_small_sizes.txt)
        u32 word[PER];
        u32 where[PER]; // run of the entry, then rank << 7 | bucket, then its position in the cell
        unsigned k = 0;    // a thread's positions grow by THREADS, about one run: the run only ever steps forward
        // all of a thread's loads go out before the first word is looked at: with the count in the same loop every iteration waited
        // for its own HBM round trip (a dozen in sequence per workgroup)
#pragma unroll
        for (unsigned j = 0; j < PER; j++) {
            const u32 p = min(tid + j * THREADS, N - 1); // positions past the end re-read the last entry and are dropped below (N > 0 here)
            while (p >= vstart[k + 1]) k++;
            where[j] = k;
            word[j] = p2[rbegin[k] + (p - vstart[k])];
        }
        K3_STAMP(2); // loads issued
#ifdef PANDA_K3_STAMPS
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): the wait for them on its own
        K3_STAMP(3);
#endif
#pragma unroll
        for (unsigned j = 0; j < PER; j++) {
            const u32 p = tid + j * THREADS;
            if (p < N) {
                const u32 v = word[j];
                word[j] = final_word(v, where[j]);
                const u32 b = v >> shift;
                where[j] = (atomicAdd(&cnt[b], 1u) << 7) | b;
            }
        }
        K3_STAMP(4); // LDS atomics
        __syncthreads();
        K3_STAMP(5);
        u32 mine = tid < 128 ? cnt[tid] : 0;
        scan128_inclusive(cnt, &scan_carry);
        if (tid < L) {
            const u32 start = cnt[tid] - mine; // local
            cur[tid] = start;
            ow[((u64)q << g.b3) + tid] = out_rel + start;
        }
        if (q == g.Q - 1 && tid == 0) ow[NB] = out_rel + N;
        __syncthreads();
#pragma unroll
        for (unsigned j = 0; j < PER; j++) {
            const u32 p = tid + j * THREADS;
            if (p < N) where[j] = cur[where[j] & 127u] + (where[j] >> 7);
        }
#pragma unroll 1
        for (u32 base = 0; base < N; base += IMG) { // one round unless PER > K3_PER
            if (base) __syncthreads(); // the image of the previous round has left
#pragma unroll
            for (unsigned j = 0; j < PER; j++) {
                const u32 p = tid + j * THREADS;
                if (p < N && where[j] - base < IMG) outbuf[where[j] - base] = word[j];
            }
            __syncthreads();
            const u32 len = min(N - base, IMG);
            for (u32 p = tid; p < len; p += THREADS) sw[out_rel + base + p] = outbuf[p];
        }
        K3_STAMP(7); // scan, offsets, ranking into the LDS image, copy out
        return;
    }

    // oversized cell (heavily skewed scalars): count, then rank straight into global memory
    for (u32 p = tid; p < N; p += THREADS) {
        const unsigned k = window_of(p);
        atomicAdd(&cnt[p2[rbegin[k] + (p - vstart[k])] >> shift], 1u);
    }
    __syncthreads();
    u32 mine = tid < 128 ? cnt[tid] : 0;
    for (unsigned d = 1; d < 128; d <<= 1) {
        u32 v = (tid < 128 && tid >= d) ? cnt[tid - d] : 0;
        __syncthreads();
        if (tid < 128) cnt[tid] += v;
        __syncthreads();
    }
    if (tid < L) {
        u32 start = out_rel + cnt[tid] - mine;
        cur[tid] = start;
        ow[((u64)q << g.b3) + tid] = start;
    }
    if (q == g.Q - 1 && tid == 0) ow[NB] = out_rel + N;
    __syncthreads();
    for (u32 p = tid; p < N; p += THREADS) {
        const unsigned k = window_of(p);
        const u32 v = p2[rbegin[k] + (p - vstart[k])];
        sw[atomicAdd(&cur[v >> shift], 1u)] = final_word(v, k);
    }
    K3_STAMP(6); // the two-pass path for oversized cells ends here (stamps 2 .. 5 and 7 stay zero)
}

// The same merge (tabled mode only) in at most 24 registers per thread: two passes over the cell's runs -- count, then rank into the LDS
// image (or, for an oversized cell, straight into the list) -- instead of sixteen words and their ranks held in registers.  A
// 1024-thread workgroup of it fits the 96 registers per SIMD that the bucket accumulation with its row staged in LDS leaves free, so the
// rest of the list can be merged BESIDE the accumulation of its front (SortSplit).  Two LDS atomics per entry instead of one.
// THREADS = 256 (a wave per SIMD, placed at once beside resident accumulate workgroups) ranks straight into the list: no 64 KB LDS image, so
// several of these workgroups fit a CU; the 4-byte stores of a cell land in its own 48 KB of the list.
template <unsigned THREADS>
__global__ void __launch_bounds__(THREADS) k3_merge_small(const u32 *__restrict__ p2, const u32 *__restrict__ part_off, const u32 *__restrict__ sub_off,
                                                 const u32 *__restrict__ cell_off, u32 *__restrict__ off, u32 *__restrict__ sorted, TabledGeom g, unsigned NB, SortRange R)
{
    __builtin_amdgcn_s_setprio(3); // beside the bucket accumulation of a split sort (SortSplit) these waves must win the issue arbitration: they issue little, it issues always
    constexpr bool STAGED = THREADS >= 1024;
    __shared__ u32 cnt[128], cur[128], scan_carry;
    __shared__ u32 outbuf[STAGED ? K3_CAP : 1];
    __shared__ u32 rbegin_lo[64], vstart[65]; // a run's start in p2 as (k << log_n) + rbegin_lo[k]
    const unsigned tid = threadIdx.x;
    const unsigned q = R.q_lo + blockIdx.x;
    const unsigned h1 = q / g.H2, h2 = q % g.H2;
    const unsigned L = 1u << g.b3;
    const u32 out_rel = cell_off[q];
    if (tid < g.W) {
        const unsigned at = seg_of(g, tid, h1) * (g.H2 + 1) + h2;
        const u32 so = sub_off[at];
        rbegin_lo[tid] = part_off[tid * (g.H1 + 1) + h1] + so;
        vstart[tid + 1] = sub_off[at + 1] - so; // lengths; prefixed below
    }
    if (tid < 128) cnt[tid] = 0;
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (unsigned k = 0; k < g.W; k++) {
            const u32 len = vstart[k + 1];
            vstart[k] = run;
            run += len;
        }
        vstart[g.W] = run;
    }
    __syncthreads();
    const u32 N = vstart[g.W]; // uniform over the block
    const unsigned shift = g.log_n + 1;
    if (q + 1 == R.q_hi && q + 1 < g.Q && tid == 0) off[(u64)(q + 1) << g.b3] = out_rel + N; // see k3_merge, closes_part
    if (N == 0) {
        if (tid < L) off[((u64)q << g.b3) + tid] = out_rel;
        if (q == g.Q - 1 && tid == 0) off[NB] = out_rel;
        return;
    }
    // pass 1: count.  A thread's positions grow by THREADS, at most about one run: the run index only ever steps forward
    unsigned k = 0;
#pragma unroll 1
    for (u32 p = tid; p < N; p += THREADS) {
        while (p >= vstart[k + 1]) k++;
        atomicAdd(&cnt[p2[((u64)k << g.log_n) + rbegin_lo[k] + (p - vstart[k])] >> shift], 1u);
    }
    __syncthreads();
    const u32 mine = tid < 128 ? cnt[tid] : 0;
    scan128_inclusive(cnt, &scan_carry);
    if (tid < L) {
        const u32 start = cnt[tid] - mine; // local
        cur[tid] = start;
        off[((u64)q << g.b3) + tid] = out_rel + start;
    }
    if (q == g.Q - 1 && tid == 0) off[NB] = out_rel + N;
    __syncthreads();
    // pass 2: the same words again (L2), each to the next free slot of its bucket
    const bool in_lds = STAGED && N <= K3_CAP;
    const u32 id_mask = (1u << g.log_n) - 1;
    k = 0;
#pragma unroll 1
    for (u32 p = tid; p < N; p += THREADS) {
        while (p >= vstart[k + 1]) k++;
        const u32 v = p2[((u64)k << g.log_n) + rbegin_lo[k] + (p - vstart[k])];
        const u32 word = ((v & id_mask) + g.row0 + (k << g.row_shift)) | (((v >> g.log_n) & 1u) << 31);
        const u32 slot = atomicAdd(&cur[v >> shift], 1u);
        if (in_lds)
            outbuf[slot] = word;
        else
            sorted[out_rel + slot] = word;
    }
    if (!in_lds) return;
    __syncthreads();
#pragma unroll 1
    for (u32 p = tid; p < N; p += THREADS) sorted[out_rel + p] = outbuf[p];
}

// ------------------------------------------------------------------------------- host side

SortGeom plain_geom(unsigned log_n, unsigned c)
{
    SortGeom g;
    g.log_n = log_n;
    g.lo_bits = std::min(std::min(7u, c - 1), 31u - log_n);
    g.H = 1u << (c - 1 - g.lo_bits);
    g.tiles = (unsigned)((((u64)1 << log_n) + SORT_TILE - 1) / SORT_TILE);
    g.row0 = 0;
    return g;
}

TabledGeom tabled_geom(unsigned log_n, const panda::WindowPlan &plan, bool per_window = false)
{
    TabledGeom g{};
    const unsigned B = plan.width[0] - 1;
    g.log_n = log_n;
    g.W = plan.W;
    g.per_window = per_window ? 1u : 0u;
    g.b3 = std::min(std::min(7u, 31u - log_n), B);
    // a level-3 cell (2^b3 buckets of all windows, or of one) should fit k3_merge's register-resident path: mean entries per cell
    // = W n 2^b3 / 2^B (n 2^b3 / 2^B per window), kept below 0.8 K3_CAP (the counts are Poisson-tight for uniform scalars).
    // The MEAN: the windows of a plan differ in width by up to a bit, and a window of c bits only reaches the lowest 2^(c-1) buckets, so
    // the cells of the lower half of the bucket space hold up to twice the mean -- 21.5 k entries at 2^24 points in 3 x 22 + 9 x 21 bits,
    // where the mean is 12.3 k.  Those cells are merged by the wide variant of k3_merge (sort3, wide_cells); until round 5 they took the
    // slow path for oversized cells and the merge ran 1.00 ms instead of 0.55 (profiles/r05_k3_merge_cells.txt).  (Sizing the cells by the
    // densest one instead -- b3 one less, 128 x 256 partitions -- costs levels 1 and 2 more than the wide variant costs level 3.)
    const u64 per_bucket_bits = per_window ? ((u64)1 << log_n) : ((u64)plan.W << log_n);
    while (g.b3 > 3 && (per_bucket_bits >> (B - g.b3)) > (u64)K3_CAP * 4 / 5) g.b3--;
    const unsigned rest = B - g.b3;
    g.b1 = (rest + 1) / 2;
    g.b2 = rest - g.b1;
    g.H1 = 1u << g.b1;
    g.H2 = 1u << g.b2;
    g.S = g.W * g.H1;
    g.Q = g.H1 * g.H2;
    const u64 E = (u64)plan.W << log_n;
    g.max_tiles2 = (unsigned)(E / SORT_TILE + g.S + 1);
    g.row_shift = log_n;
    g.row0 = 0;
    return g;
}


template <class Fr, class Code>
void launch_digits(hipStream_t stream, const void *scalars, Code *dig, u64 n, const panda::WindowPlan &plan, const panda::SampleCheck &sc)
{
    hipLaunchKernelGGL((k_digits<Fr, Code>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const u32 *)scalars, dig, n, plan, sc);
}

template <class Code>
void launch_digits_for(unsigned fr, hipStream_t stream, const void *scalars, Code *dig, u64 n, const panda::WindowPlan &plan, const panda::SampleCheck &sc)
{
    switch (fr) {
    case 0: launch_digits<Bn254Fr, Code>(stream, scalars, dig, n, plan, sc); break;
    case 1: launch_digits<Bls377Fr, Code>(stream, scalars, dig, n, plan, sc); break;
    default: launch_digits<Bls381Fr, Code>(stream, scalars, dig, n, plan, sc); break;
    }
}

// digits + level-1 histogram; returns false if the fused kernel does not apply (the caller then runs k_digits + k_part_hist)
template <class Fr, class Code>
bool launch_digits_hist(hipStream_t stream, const void *scalars, Code *dig, u32 *tile_hist, u64 n, const panda::WindowPlan &plan, const SortGeom &g,
                        const panda::SampleCheck &sc)
{
    const size_t lds = (size_t)plan.W * g.H * 4;
    if (g.tiles < 512 || lds > 48 * 1024) return false;
    const DigitsHistGeom dg{g.lo_bits, g.H, g.tiles};
    // a thread walks its scalars one after the other (load, convert, W stores): with 256 threads per tile a 2^22-point call ran two waves
    // per SIMD and the kernel was bound by those round trips (0.23 ms for 0.35 GB); 1024 threads until the tiles alone fill the chip
    const unsigned threads = g.tiles >= 8192 ? 512u : 1024u;
    hipLaunchKernelGGL((k_digits_hist<Fr, Code>), dim3(g.tiles), dim3(threads), lds, stream, (const u32 *)scalars, dig, tile_hist, n, plan, dg, sc);
    return true;
}

template <class Code>
bool launch_digits_hist_for(unsigned fr, hipStream_t stream, const void *scalars, Code *dig, u32 *tile_hist, u64 n, const panda::WindowPlan &plan,
                            const SortGeom &g, const panda::SampleCheck &sc)
{
    switch (fr) {
    case 0: return launch_digits_hist<Bn254Fr, Code>(stream, scalars, dig, tile_hist, n, plan, g, sc);
    case 1: return launch_digits_hist<Bls377Fr, Code>(stream, scalars, dig, tile_hist, n, plan, g, sc);
    default: return launch_digits_hist<Bls381Fr, Code>(stream, scalars, dig, tile_hist, n, plan, g, sc);
    }
}

} // namespace

namespace panda {

WindowPlan make_window_plan(unsigned total_bits, unsigned c)
{
    WindowPlan p{};
    if (c < 4) c = 4; // narrower windows would not fit MAX_WINDOWS
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

template <class Fr>
static WindowPlan safe_plan(unsigned c)
{
    // BITS + 1: one spare bit for the signed-digit carry.  The top window must never carry out: its largest raw value
    // (top bits of r - 1, plus a carry in) has to stay below half its range; widen the plan by a bit until it does.
    constexpr int LR = Fr::L;
    WindowPlan plan{};
    for (unsigned total = Fr::BITS + 1;; total++) {
        plan = make_window_plan(total, c);
        const unsigned lo = plan.lo[plan.W - 1], m = lo >> 5, sh = lo & 31;
        u64 top = m < (unsigned)LR ? ((u64)Fr::PW[m] >> sh) : 0;
        if (m + 1 < (unsigned)LR) top |= (u64)Fr::PW[m + 1] << (32 - sh);
        if (m + 2 < (unsigned)LR && sh) top |= (u64)Fr::PW[m + 2] << (64 - sh);
        if (top + 1 < ((u64)1 << (plan.width[plan.W - 1] - 1)) || total > Fr::BITS + 8) break;
    }
    return plan;
}

WindowPlan make_safe_window_plan(unsigned fr, unsigned c) { return fr == 0 ? safe_plan<Bn254Fr>(c) : (fr == 1 ? safe_plan<Bls377Fr>(c) : safe_plan<Bls381Fr>(c)); }

static bool sort3_supported(unsigned log_n, const WindowPlan &plan, bool per_window);
static size_t sort3_bytes(unsigned log_n, const WindowPlan &plan, bool per_window);
static hipError_t sort3(hipStream_t stream, Arena &arena, unsigned fr, const void *scalars, unsigned log_n, const WindowPlan &plan, SortEvents ev, SortResult *out,
                        SortPlacement place, SampleCheck check, bool per_window, SortSplit *split = nullptr);

bool msm_sort_plain_supported(unsigned log_n, const WindowPlan &plan)
{
    const unsigned c = plan.width[0];
    if (c <= 16) return c >= 2 && plain_geom(log_n, c).H <= MAX_PARTS;
    return sort3_supported(log_n, plan, true);
}

size_t msm_sort_plain_bytes(unsigned log_n, const WindowPlan &plan)
{
    if (plan.width[0] > 16) return sort3_bytes(log_n, plan, true);
    const u64 n = (u64)1 << log_n;
    const unsigned W = plan.W, c = plan.width[0], NB = 1u << (c - 1);
    const SortGeom g = plain_geom(log_n, c);
    return align256(n * W * 2 + 16) + 2 * align256((size_t)W * g.tiles * g.H * 4) + 2 * align256((size_t)W * (g.H + 1) * 4) + align256((size_t)W * (NB + 1) * 4) +
           2 * align256(n * W * 4) + 4096;
}

hipError_t msm_sort_plain(hipStream_t stream, Arena &arena, unsigned fr, const void *scalars, unsigned log_n, const WindowPlan &plan, SortEvents ev,
                          SortResult *out, SortPlacement place, SampleCheck check)
{
    // windows wider than 16 bits (u32 digit codes, 2^16 .. 2^19 buckets per window) take the three-level sort, a list per window
    if (plan.width[0] > 16) return sort3(stream, arena, fr, scalars, log_n, plan, ev, out, place, check, true);
    const u64 n = (u64)1 << log_n;
    const unsigned W = plan.W, c = plan.width[0], NB = 1u << (c - 1);
    if (c < 2) return hipErrorInvalidValue;
    SortGeom geom = plain_geom(log_n, c);
    if (geom.H > MAX_PARTS) return hipErrorInvalidValue;
    if (place.row_shift) {
        if ((u64)place.row0 + n > ((u64)1 << 31)) return hipErrorInvalidValue;
        geom.row0 = place.row0;
    }
    uint16_t *d_dig = (uint16_t *)arena.take(n * W * 2 + 16);
    u32 *d_thist = (u32 *)arena.take((size_t)W * geom.tiles * geom.H * 4);
    u32 *d_tpref = (u32 *)arena.take((size_t)W * geom.tiles * geom.H * 4);
    u32 *d_poff = (u32 *)arena.take((size_t)W * (geom.H + 1) * 4);
    u32 *d_ptot = (u32 *)arena.take((size_t)W * (geom.H + 1) * 4);
    u32 *d_p1 = (u32 *)arena.take(n * W * 4);
    u32 *d_off = (u32 *)arena.take((size_t)W * (NB + 1) * 4);
    u32 *d_sorted = (u32 *)arena.take(n * W * 4);
    if (!d_dig || !d_thist || !d_tpref || !d_poff || !d_ptot || !d_p1 || !d_off || !d_sorted) return hipErrorOutOfMemory;

    const bool fused = launch_digits_hist_for<uint16_t>(fr, stream, scalars, d_dig, d_thist, n, plan, geom, check);
    if (!fused) launch_digits_for<uint16_t>(fr, stream, scalars, d_dig, n, plan, check);
    if (ev.digits_done) PANDA_TRY(hipEventRecord(ev.digits_done, stream));
    if (!fused) hipLaunchKernelGGL(k_part_hist<uint16_t>, dim3(geom.tiles, W), dim3(256), 0, stream, d_dig, d_thist, geom);
    hipLaunchKernelGGL(k_part_scan_cols, dim3((geom.H + 15) / 16, W), dim3(1024), 0, stream, d_thist, d_tpref, d_ptot, geom);
    hipLaunchKernelGGL(k_part_offsets, dim3(W), dim3(1024), 0, stream, d_ptot, d_poff, geom);
    hipLaunchKernelGGL((k_part_scatter<uint16_t, u32>), dim3(geom.tiles, W), dim3(SORT_THREADS), 0, stream, d_dig, d_thist, d_tpref, d_poff, d_p1, geom);
    if (ev.partition_done) PANDA_TRY(hipEventRecord(ev.partition_done, stream));
    // (writing the ranked words straight to their slots, without the LDS staging, measured 1.5x slower)
    hipLaunchKernelGGL(k_bucket_sort, dim3(geom.H, W), dim3(SORT_THREADS), 0, stream, d_p1, d_poff, d_off, d_sorted, geom, NB);
    out->off = d_off;
    out->sorted = d_sorted;
    out->lists = W;
    out->NB = NB;
    out->stride = n;
    return hipGetLastError();
}

static bool sort3_supported(unsigned log_n, const WindowPlan &plan, bool per_window)
{
    const unsigned c = plan.width[0];
    if (plan.W > 32 || plan.W < 1 || c < 4 || c > 24 || log_n > 26 || log_n < 1) return false;
    unsigned wbits = 0;
    while ((1u << wbits) < plan.W) wbits++;
    if (log_n + wbits > 31) return false;
    const TabledGeom g = tabled_geom(log_n, plan, per_window);
    return g.b1 <= 10 && g.b2 <= 8 && g.b3 <= 7 && g.S <= 16384 && g.b3 + 1 + log_n <= 32;
}

static size_t sort3_bytes(unsigned log_n, const WindowPlan &plan, bool per_window)
{
    const u64 E = (u64)plan.W << log_n;
    const unsigned NB = 1u << (plan.width[0] - 1);
    const TabledGeom g = tabled_geom(log_n, plan, per_window);
    const unsigned tiles1 = (unsigned)((((u64)1 << log_n) + SORT_TILE - 1) / SORT_TILE);
    const size_t cells = (size_t)g.Q * (per_window ? g.W : 1u), lists = per_window ? g.W : 1u;
    return align256(E * 4 + 16) + 2 * align256((size_t)g.W * tiles1 * g.H1 * 4) + 2 * align256((size_t)g.W * (g.H1 + 1) * 4) + align256(E * 4 + 64) + align256(E + 16) +
           align256((size_t)(g.S + 1) * 4) + 2 * align256((size_t)g.max_tiles2 * g.H2 * 4) + align256((size_t)g.S * g.H2 * 4) +
           align256((size_t)g.S * (g.H2 + 1) * 4) + align256(E * 4) + 2 * align256((cells + 1) * 4) + align256((cells / 1024 + 1) * 4) +
           align256(lists * (NB + 1) * 4) + align256(E * 4) + 8192;
}

static std::atomic<unsigned> g_wide_merge{0};

// the three-level sort: tabled mode (one list over all windows) or per-window mode (W lists)
static hipError_t sort3(hipStream_t stream, Arena &arena, unsigned fr, const void *scalars, unsigned log_n, const WindowPlan &plan, SortEvents ev, SortResult *out,
                        SortPlacement place, SampleCheck check, bool per_window, SortSplit *split)
{
    if (!sort3_supported(log_n, plan, per_window)) return hipErrorInvalidValue;
    const u64 n = (u64)1 << log_n;
    const u64 E = (u64)plan.W << log_n;
    const unsigned W = plan.W, NB = 1u << (plan.width[0] - 1);
    TabledGeom g = tabled_geom(log_n, plan, per_window);
    if (place.row_shift) { // a point range [row0, row0 + n) of tables that hold 2^row_shift rows each (per-window mode: of the base array)
        if (place.row_shift < log_n || place.row_shift > 26) return hipErrorInvalidValue;
        const u64 last_row = (per_window ? 0 : ((u64)(W - 1) << place.row_shift)) + place.row0 + n;
        if (last_row > ((u64)1 << 31)) return hipErrorInvalidValue;
        g.row_shift = place.row_shift;
        g.row0 = place.row0;
    }
    SortGeom g1;
    g1.row0 = 0;
    g1.log_n = log_n;
    g1.lo_bits = g.b2 + g.b3;
    g1.H = g.H1;
    g1.tiles = (unsigned)((n + SORT_TILE - 1) / SORT_TILE);
    const unsigned cells = g.Q * (per_window ? W : 1u), lists = per_window ? W : 1u;
    const unsigned qblocks = (cells + 1023) / 1024;

    u32 *d_dig = (u32 *)arena.take(E * 4 + 16);
    u32 *d_thist1 = (u32 *)arena.take((size_t)W * g1.tiles * g.H1 * 4);
    u32 *d_tpref1 = (u32 *)arena.take((size_t)W * g1.tiles * g.H1 * 4);
    u32 *d_poff = (u32 *)arena.take((size_t)W * (g.H1 + 1) * 4);
    u32 *d_ptot = (u32 *)arena.take((size_t)W * (g.H1 + 1) * 4);
    u32 *d_p1_lo = (u32 *)arena.take(E * 4 + 64);
    unsigned char *d_p1_hi = (unsigned char *)arena.take(E + 16);
    u32 *d_segtile = (u32 *)arena.take((size_t)(g.S + 1) * 4);
    u32 *d_thist2 = (u32 *)arena.take((size_t)g.max_tiles2 * g.H2 * 4);
    u32 *d_tpref2 = (u32 *)arena.take((size_t)g.max_tiles2 * g.H2 * 4);
    u32 *d_tot2 = (u32 *)arena.take((size_t)g.S * g.H2 * 4);
    u32 *d_suboff = (u32 *)arena.take((size_t)g.S * (g.H2 + 1) * 4);
    u32 *d_p2 = (u32 *)arena.take(E * 4);
    u32 *d_cellcnt = (u32 *)arena.take((size_t)cells * 4);
    u32 *d_blksum = (u32 *)arena.take((size_t)qblocks * 4);
    u32 *d_celloff = (u32 *)arena.take((size_t)(cells + 1) * 4);
    u32 *d_off = (u32 *)arena.take((size_t)lists * (NB + 1) * 4);
    u32 *d_sorted = (u32 *)arena.take(E * 4);
    if (!d_dig || !d_thist1 || !d_tpref1 || !d_tpref2 || !d_poff || !d_ptot || !d_p1_lo || !d_p1_hi || !d_segtile || !d_thist2 || !d_tot2 || !d_suboff || !d_p2 || !d_cellcnt || !d_blksum || !d_celloff ||
        !d_off || !d_sorted)
        return hipErrorOutOfMemory;

    const bool fused = launch_digits_hist_for<u32>(fr, stream, scalars, d_dig, d_thist1, n, plan, g1, check);
    if (!fused) launch_digits_for<u32>(fr, stream, scalars, d_dig, n, plan, check);
    if (ev.digits_done) PANDA_TRY(hipEventRecord(ev.digits_done, stream));
    // level 1, per window
    if (!fused) hipLaunchKernelGGL(k_part_hist<u32>, dim3(g1.tiles, W), dim3(256), 0, stream, d_dig, d_thist1, g1);
    hipLaunchKernelGGL(k_part_scan_cols, dim3((g1.H + 15) / 16, W), dim3(1024), 0, stream, d_thist1, d_tpref1, d_ptot, g1);
    hipLaunchKernelGGL(k_part_offsets, dim3(W), dim3(1024), 0, stream, d_ptot, d_poff, g1);
    hipLaunchKernelGGL(k1_scatter_split<u32>, dim3(g1.tiles, W), dim3(SORT_THREADS), 0, stream, d_dig, d_thist1, d_tpref1, d_poff, d_p1_lo, d_p1_hi, g1, g.b3);
    hipLaunchKernelGGL(k2_seg_tiles, dim3(1), dim3(SORT_THREADS), 0, stream, d_poff, d_segtile, g);
    // Levels 2 and 3 for the level-1 partitions [h1_lo, h1_hi) -- a contiguous range of segments, tiles, cells and buckets -- on stream s.
    // The launch bound of the tile kernels is the whole list's (how the entries spread over the partitions is the scalars' business);
    // workgroups beyond the range's last tile leave after two loads.
    auto level2 = [&](hipStream_t s, const SortRange &R) {
        hipLaunchKernelGGL(k2_hist, dim3(g.max_tiles2), dim3(256), 0, s, d_p1_hi, d_poff, d_segtile, d_thist2, g, R);
        hipLaunchKernelGGL(k2_scan_cols, dim3((g.H2 + 3) / 4, R.s_hi - R.s_lo), dim3(256), 0, s, d_thist2, d_tpref2, d_segtile, d_tot2, g, R);
        hipLaunchKernelGGL(k2_offsets, dim3(R.s_hi - R.s_lo), dim3(256), 0, s, d_tot2, d_suboff, g, R);
        if (s == stream)
            hipLaunchKernelGGL(k2_scatter<SORT_THREADS>, dim3(g.max_tiles2), dim3(SORT_THREADS), 0, s, d_p1_lo, d_p1_hi, d_poff, d_segtile, d_thist2, d_tpref2, d_suboff, d_p2, g, R);
        else
            hipLaunchKernelGGL(k2_scatter<256>, dim3(g.max_tiles2), dim3(256), 0, s, d_p1_lo, d_p1_hi, d_poff, d_segtile, d_thist2, d_tpref2, d_suboff, d_p2, g, R);
    };
    auto level3_counts = [&](hipStream_t s, const SortRange &R) {
        hipLaunchKernelGGL(k3_cell_counts, dim3((R.q_hi - R.q_lo + 1023) / 1024), dim3(s == stream ? 1024 : 256), 0, s, d_suboff, d_cellcnt, d_blksum, g, R);
    };
    // cells [0, wide_cells) hold more than the ordinary merge reads at once (and at most what the wide one does): in a plan of two widths the
    // cells of the lower half of the bucket space, which every window reaches
    unsigned wide_cells = 0, small_from = ~0u; // cells [small_from, Q): few enough entries for 256-thread workgroups
    if (!per_window) {
        double per_bucket = 0;
        unsigned narrow = plan.width[0];
        for (unsigned k = 0; k < plan.W; k++) {
            per_bucket += ldexp((double)n, -(int)(plan.width[k] - 1));
            narrow = std::min(narrow, (unsigned)plan.width[k]);
        }
        const double densest = ldexp(per_bucket, (int)g.b3);
        if (narrow + 1 == plan.width[0] && densest > (double)K3_CAP * 0.8 && densest <= (double)K3_THREADS * K3_PER_WIDE * 0.8) wide_cells = g.Q / 2;
        // the other half of a two-width plan: only the wide windows reach it
        double sparse_bucket = 0;
        for (unsigned k = 0; k < plan.W; k++)
            if (plan.width[k] == plan.width[0]) sparse_bucket += ldexp((double)n, -(int)(plan.width[k] - 1));
        if (narrow + 1 == plan.width[0] && ldexp(sparse_bucket, (int)g.b3) <= 256.0 * K3_PER * 0.85) small_from = g.Q / 2;
        const unsigned mode = g_wide_merge.load(std::memory_order_relaxed); // tests: every cell through the wide variant, or none
        if (mode == 1) wide_cells = g.Q, small_from = ~0u;
        if (mode == 2) wide_cells = 0, small_from = ~0u;
        if (mode == 3) wide_cells = 0, small_from = 0; // every cell through the 256-thread variant
    }
    auto level3_merge = [&](hipStream_t s, const SortRange &R) {
        hipLaunchKernelGGL(k3_cell_offsets, dim3((R.q_hi - R.q_lo + 1023) / 1024), dim3(s == stream ? 1024 : 256), 0, s, d_cellcnt, d_blksum, d_celloff, g, R);
        if (s == stream) {
            // the dense half of the cells (tabled_geom) with the wide variant where they would not fit the ordinary one
            const unsigned wide_hi = std::min(std::max(wide_cells, R.q_lo), R.q_hi);
            if (wide_hi > R.q_lo) {
                const SortRange Rw{R.s_lo, R.s_hi, R.q_lo, wide_hi};
                hipLaunchKernelGGL(k3_merge<K3_PER_WIDE>, dim3(wide_hi - R.q_lo), dim3(K3_THREADS), 0, s, d_p2, d_poff, d_suboff, d_celloff, d_off, d_sorted, g, NB, Rw);
            }
            const unsigned small_lo = std::min(std::max(small_from, wide_hi), R.q_hi);
            if (small_lo > wide_hi) {
                const SortRange Rn{R.s_lo, R.s_hi, wide_hi, small_lo};
                hipLaunchKernelGGL(k3_merge<K3_PER>, dim3(small_lo - wide_hi), dim3(K3_THREADS), 0, s, d_p2, d_poff, d_suboff, d_celloff, d_off, d_sorted, g, NB, Rn);
            }
            if (R.q_hi > small_lo) {
                const SortRange Rs{R.s_lo, R.s_hi, small_lo, R.q_hi};
                hipLaunchKernelGGL((k3_merge<K3_PER, 256>), dim3(R.q_hi - small_lo), dim3(256), 0, s, d_p2, d_poff, d_suboff, d_celloff, d_off, d_sorted, g, NB, Rs);
            }
        } else // on the helper stream, beside the accumulation of the front: the variant that fits next to it
            hipLaunchKernelGGL(k3_merge_small<256>, dim3(R.q_hi - R.q_lo), dim3(256), 0, s, d_p2, d_poff, d_suboff, d_celloff, d_off, d_sorted, g, NB, R);
    };
    // the cut between the front and the rest: a level-1 partition boundary whose first cell starts a block of 1024 cells
    unsigned cut_h1 = 0;
    if (split) {
        split->active = false;
        const unsigned step = g.H2 >= 1024 ? 1u : 1024u / g.H2; // partitions per block of cells
        if (split->helper && !per_window && split->front_of_128 && g.H1 >= 2 * step && g.Q % 1024 == 0) {
            cut_h1 = (unsigned)(((u64)g.H1 * split->front_of_128 / 128 + step / 2) / step * step);
            cut_h1 = std::min(std::max(cut_h1, step), g.H1 - step);
        }
    }
    if (cut_h1) {
        const SortRange front{0, cut_h1 * W, 0, cut_h1 * g.H2}, rest{cut_h1 * W, g.S, cut_h1 * g.H2, g.Q};
        level2(stream, front);
        if (ev.partition_done) PANDA_TRY(hipEventRecord(ev.partition_done, stream));
        level3_counts(stream, front);
        level3_merge(stream, front);
        // the rest starts when the front is sorted, beside whatever the caller enqueues next (started earlier it shares the chip with the
        // front's own levels 2 and 3, which are on the critical path: 1.6 ms instead of 0.4 at 2^24)
        PANDA_TRY(hipEventRecord(split->ev_front, stream));
        PANDA_TRY(hipStreamWaitEvent(split->helper, split->ev_front, 0));
        level2(split->helper, rest);
        level3_counts(split->helper, rest); // its cell offsets continue the front's block sums (complete: ev_front)
        level3_merge(split->helper, rest);
        PANDA_TRY(hipEventRecord(split->rest_done, split->helper));
        split->active = true;
        split->pos = d_celloff;
        split->cells = g.Q;
        split->cut_cell = cut_h1 * g.H2;
        split->cut_bucket = split->cut_cell << g.b3;
    } else {
        const SortRange all{0, g.S, 0, cells};
        level2(stream, all);
        if (ev.partition_done) PANDA_TRY(hipEventRecord(ev.partition_done, stream));
        level3_counts(stream, all);
        level3_merge(stream, all);
    }
    out->off = d_off;
    out->sorted = d_sorted;
    out->lists = lists;
    out->NB = NB;
    out->stride = per_window ? n : E;
    return hipGetLastError();
}

#ifdef PANDA_K3_STAMPS
extern "C" int panda_debug_k3_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k3_stamps), sizeof(unsigned long long) * 1024 * 8); }
#endif
void msm_sort_set_wide_merge(unsigned mode) { g_wide_merge.store(mode, std::memory_order_relaxed); }
bool msm_sort_tabled_supported(unsigned log_n, const WindowPlan &plan) { return sort3_supported(log_n, plan, false); }
size_t msm_sort_tabled_bytes(unsigned log_n, const WindowPlan &plan) { return sort3_bytes(log_n, plan, false); }

hipError_t msm_sort_tabled(hipStream_t stream, Arena &arena, unsigned fr, const void *scalars, unsigned log_n, const WindowPlan &plan, SortEvents ev,
                           SortResult *out, SortPlacement place, SampleCheck check, SortSplit *split)
{
    return sort3(stream, arena, fr, scalars, log_n, plan, ev, out, place, check, false, split);
}

} // namespace panda
