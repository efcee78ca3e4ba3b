// ntt.hip -- radix-2 Cooley-Tukey number-theoretic transform over BN254 Fr for gfx950.
//
// Drop-in for panda_ntt_execute_bn254[_v1] (src/cuda/core/unit/ntt/fft.cu:171-260).  The reference ships the
// driver only -- its kernel bodies are compiled out (fft.cu:18-35,89-101,117-168) -- so what is kept is
//   * the transform the commented-out code defines: natural order in, natural order out,
//     y[k] = sum_j x[j] w^(jk), Montgomery-form residues, no 1/n scaling;
//   * the buffer protocol: ceil(log_n / 8) passes of radix 2^deg (deg = min(8, remaining), fft.cu:177,193-210),
//     ping-pong between d_src and d_dst, *flag = passes & 1 tells the caller where the answer is (fft.cu:211,
//     unit.rs:521-532).
// and everything else is new.  Since round 3 the passes of transforms of 2^11 points and more live in ntt_radix8.h (k_ntt_pass8:
// eight elements per thread in registers, three register blocks with two LDS exchanges, precomputed-quotient twiddle products, the
// twiddle between passes on the output side; k_ntt_small: radix 2 / 4 / 8 passes without LDS).  This file keeps the host side -- tables,
// pass loop, flag protocol, slab steps, C entry points -- and k_ntt_pass, the kernel of rounds 1-2, for transforms below 2^11 points:
//   * one workgroup = one 1024-element tile (4 radix-256 sub-transforms) held in LDS as 9 limb planes, so a
//     butterfly's 18 ds_read_b32 / ds_write_b32 are bank-conflict-free across the wave;
//   * the residues stay in the caller's Montgomery radix (2^256): multiplying by a twiddle held in the
//     kernels' own radix (2^261, fe29.h) maps wire form to wire form, so no conversion pass exists;
//   * butterfly twiddles (128 per pass) are staged in LDS; the inter-pass twiddles w^(e k i) come from two
//     L2-resident power tables (<= 2^16 and <= 2^12 entries) and one multiply, never from an n-entry table;
//   * sums are never reduced inside a pass: bounds are tracked at compile time and one multiply-free
//     reduction (fe_reduce_small) is inserted where the lazy headroom (R/p = 169) would run out;
//   * the inverse transform folds n^-1 into the last pass's twiddle table.
//
// Multi-GPU (no reference counterpart): n = G m over G ranks.  Rank r holds the decimated slab
// X_r[j2] = x[r + G j2].  step1 = local m-point transform with root w^G, then the twiddle w^(r k2);
// the caller's all-to-all moves chunk q (k2 in [q m/G, (q+1) m/G)) to rank q; step2 = m/G transforms of
// size G with root w^m down the received pieces.  Rank q ends with y[k1 m + q m/G + k2'] at [k1][k2'].
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string.h>

#include "fe29.h"
#include "ntt_radix8.h"
#include "ntt_radix9.h"
#include "panda_internal.h"

using namespace panda29;

namespace {

constexpr int NL = 9;           // limbs of every supported scalar field (BN254 Fr, BLS12-377 Fr)
constexpr int TW_STRIDE = 12;   // table entries padded to 48 B for 16-byte loads
constexpr int TILE = 1024;      // elements per workgroup

constexpr int POW_BITS = 26; // exponents of the power tables stay below 2^26 (the streamed inter-pass table of a 2^26-point transform: 2 GiB)
struct PowBase {
    u32 pw[POW_BITS][NL]; // base^(2^j), canonical internal form
    u32 scale[NL];  // optional factor folded into every entry
    int has_scale;
};

// out[t] = base^e(t) (* scale), canonical, t < count.  e(t) = t, or, for the two-dimensional tables of the wide inter-pass
// twiddles (rows_deg != 0), e(t) = (t >> rows_deg) * (t & (2^rows_deg - 1)): row a holds base^(a i), i < 2^rows_deg.  e < 2^24.
template <class Fr>
__global__ void __launch_bounds__(256) k_pow_table(PowBase pb, unsigned count, unsigned rows_deg, u32 *__restrict__ out)
{
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const unsigned t = rows_deg ? (idx >> rows_deg) * (idx & ((1u << rows_deg) - 1)) : idx;
    Fe<Fr> acc, f;
    if (pb.has_scale) {
#pragma unroll
        for (int i = 0; i < NL; i++) acc.l[i] = pb.scale[i];
    } else
        fe_one(acc);
#pragma unroll
    for (int j = 0; j < POW_BITS; j++) {
        if ((t >> j) & 1) {
#pragma unroll
            for (int i = 0; i < NL; i++) f.l[i] = pb.pw[j][i];
            fe_mul(acc, acc, f);
        }
    }
    fe_reduce_once(acc); // ONE and products are < 2p tight
    u32 *dst = out + (size_t)idx * TW_STRIDE;
#pragma unroll
    for (int i = 0; i < NL; i++) dst[i] = acc.l[i];
#pragma unroll
    for (int i = NL; i < TW_STRIDE; i++) dst[i] = 0;
}

// the same powers as (w, floor(w R / p)) pairs for the precomputed-quotient products of k_ntt_pass8 (ntt_radix8.h)
template <class Fr>
__global__ void __launch_bounds__(256) k_pow_table2(PowBase pb, unsigned count, unsigned rows_deg, u32 *__restrict__ out)
{
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const unsigned t = rows_deg ? (idx >> rows_deg) * (idx & ((1u << rows_deg) - 1)) : idx;
    Fe<Fr> acc, f;
    if (pb.has_scale) {
#pragma unroll
        for (int i = 0; i < NL; i++) acc.l[i] = pb.scale[i];
    } else
        fe_one(acc);
#pragma unroll
    for (int j = 0; j < POW_BITS; j++) {
        if ((t >> j) & 1) {
#pragma unroll
            for (int i = 0; i < NL; i++) f.l[i] = pb.pw[j][i];
            fe_mul(acc, acc, f);
        }
    }
    FeTw<Fr> tw;
    fe_shoup_prepare(tw, acc);
    u32 *dst = out + (size_t)idx * panda_ntt8::TW2_STRIDE;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        dst[i] = tw.w[i];
        dst[NL + i] = tw.q[i];
    }
    dst[2 * NL] = dst[2 * NL + 1] = 0;
}

// the same powers as 32-byte wire elements in the kernels' own Montgomery radix (w 2^261 mod p, canonical, packed): the streamed table of
// the middle pass's single output product (Pass8Args::wide)
template <class Fr>
__global__ void __launch_bounds__(256) k_pow_table_wire(PowBase pb, unsigned count, unsigned rows_deg, u32 *__restrict__ out)
{
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const unsigned t = (idx >> rows_deg) * (idx & ((1u << rows_deg) - 1));
    Fe<Fr> acc, f;
    fe_one(acc);
#pragma unroll
    for (int j = 0; j < POW_BITS; j++) {
        if ((t >> j) & 1) {
#pragma unroll
            for (int i = 0; i < NL; i++) f.l[i] = pb.pw[j][i];
            fe_mul(acc, acc, f);
        }
    }
    fe_reduce_once(acc);
    u32 w8[8];
    fe_pack(w8, acc);
    uint4 *d4 = reinterpret_cast<uint4 *>(out + (size_t)idx * 8);
    d4[0] = make_uint4(w8[0], w8[1], w8[2], w8[3]);
    d4[1] = make_uint4(w8[4], w8[5], w8[6], w8[7]);
}

template <class Fr>
__device__ __forceinline__ void load_tw(Fe<Fr> &r, const u32 *__restrict__ tab, unsigned idx)
{
    const uint4 *s = reinterpret_cast<const uint4 *>(tab + (size_t)idx * TW_STRIDE);
    uint4 a = s[0], b = s[1], c = s[2];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x;
}

template <class Fr>
__device__ __forceinline__ void load_elem(Fe<Fr> &v, const u32 *__restrict__ src)
{
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 lo = s4[0], hi = s4[1];
    u32 w8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    fe_unpack(v, w8);
}

// canonical element -> 32 bytes
template <class Fr>
__device__ __forceinline__ void store_elem(u32 *__restrict__ dst, const Fe<Fr> &v)
{
    u32 w8[8];
    fe_pack(w8, v);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    d4[0] = make_uint4(w8[0], w8[1], w8[2], w8[3]);
    d4[1] = make_uint4(w8[4], w8[5], w8[6], w8[7]);
}

__device__ __forceinline__ unsigned bitrev(unsigned v, unsigned bits) { return __brev(v) >> (32 - bits); } // bits >= 1
__device__ __forceinline__ unsigned bitrev0(unsigned v, unsigned bits) { return bits ? __brev(v) >> (32 - bits) : 0u; }

// LDS image of field elements as NL limb planes: element e lives at word pad(e) = e + (e >> 5) of each plane.
// The one-word pad per 32 elements makes every access pattern of a pass conflict-free or 2-way at worst:
// unit stride (butterflies with bit >= 32), stride 2 (last rounds), stride 256 (the gather from HBM puts consecutive
// lanes into different sub-transforms) and the bit-reversed read-out.
template <int COUNT>
struct Planes {
    static constexpr int STRIDE = COUNT + COUNT / 32;
    u32 *base;
    __device__ __forceinline__ static unsigned pad(unsigned e) { return e + (e >> 5); }
    template <class Fr>
    __device__ __forceinline__ void load(Fe<Fr> &r, unsigned e) const
    {
        const unsigned p = pad(e);
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = base[i * STRIDE + p];
    }
    template <class Fr>
    __device__ __forceinline__ void store(const Fe<Fr> &r, unsigned e) const
    {
        const unsigned p = pad(e);
#pragma unroll
        for (int i = 0; i < NL; i++) base[i * STRIDE + p] = r.l[i];
    }
};
typedef Planes<TILE> TilePlanes;
typedef Planes<128> TwiddlePlanes;

// Rounds RND..DEG-1 of the radix-2^DEG sub-transform; BC = bound (units of p) of every element in LDS.
// A butterfly stores s = a + b (< 2 BC p) and d = (a - b + (BC+1) p) * w (< 2p out of the product; un-multiplied, < (2 BC + 1) p,
// in the last round).  The product needs (2 BC + 1) < 0.9 R/p at every round; when the NEXT round would break that, this round
// brings its sum below p with the multiply-free fe_reduce_small before storing it -- one reduction per butterfly, on the sum only
// (the difference is a fresh product), once every five or so rounds.
template <class Fr, int DEG, int RND, int BC>
struct Rounds {
    static constexpr long long LIM = Fr::HEADROOM * 9 / 10;
    static_assert(2 * BC + 1 < LIM, "(a - b + (BC+1) p) * twiddle must stay below 0.9 R p");
    static constexpr bool LAST = RND == DEG - 1;
    static constexpr bool REDUCE_SUM = !LAST && (2 * (2 * BC) + 1) >= LIM;
    static constexpr int NEXT = LAST ? 2 * BC + 1 : (REDUCE_SUM ? 2 : 2 * BC);
    static constexpr int FINAL = Rounds<Fr, DEG, RND + 1, NEXT>::FINAL;
    // Rounds whose butterflies fall into at most eight twiddle classes (bit <= 8: the last rounds but one of a 256-point sub-transform)
    // hand the butterflies of a full tile out BY CLASS: a wave then holds one twiddle only, and the waves of class 0 -- twiddle 1: half
    // of round 6, a quarter of round 5, an eighth of round 4, 0.44 multiplications per element and pass -- replace the product by a
    // multiply-free reduction of the difference.  The limb-plane padding keeps the strided accesses of this mapping conflict-free
    // (element stride 2 bit: banks 4 (j mod 8) + j / 8 and the like are distinct over a half-wave).
    static constexpr bool BY_CLASS = DEG == 8 && !LAST && ((1u << (DEG - 1)) >> RND) <= 8;
    __device__ __forceinline__ static void run(const TilePlanes &u, const TwiddlePlanes &pq, unsigned blk_base, unsigned t, unsigned tid, bool full_tile)
    {
        constexpr unsigned R = 1u << DEG;
        constexpr unsigned bit = (R >> 1) >> RND;
        unsigned base = blk_base, tt = t;
        bool unit_twiddle = false; // wave-uniform
        if constexpr (BY_CLASS) {
            if (full_tile) { // 4 sub-transforms x 128 butterflies over 8 waves
                constexpr unsigned WPC = 8 / bit;       // waves per class
                constexpr unsigned PER_SUB = 128 / bit; // butterflies of one class in one sub-transform
                const unsigned w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
                const unsigned cls = w / WPC;
                const unsigned c = (w % WPC) * 64 + lane;
                base = (c / PER_SUB) * R;
                tt = (c % PER_SUB) * bit + cls;
                unit_twiddle = cls == 0;
            }
        }
        const unsigned di = tt & (bit - 1);
        const unsigned i0 = (tt << 1) - di, i1 = i0 + bit;
        Fe<Fr> a, b, s, d;
        u.load(a, base + i0);
        u.load(b, base + i1);
        if (REDUCE_SUM) {
            fe_add_nr(s, a, b); // limbs < 2^31: fe_reduce_small carries first
            fe_reduce_small_2p(s); // < 2p, like the products next to it
        } else if (LAST)
            fe_add_nr(s, a, b); // the read-out below carries before it reduces: no normalisation here
        else
            fe_add(s, a, b);
        if (LAST) { // last round: every twiddle is 1; limbs < 2^32 are fine for the read-out
            constexpr int K = BC + SubMargin<Fr>::value;
#pragma unroll
            for (int i = 0; i < NL; i++) d.l[i] = a.l[i] + Fr::KP[K][i] - b.l[i];
        } else if (BY_CLASS && unit_twiddle) { // the whole wave multiplies by 1: a - b + K p, brought below 2p without a product
            constexpr int K = BC + SubMargin<Fr>::value;
#pragma unroll
            for (int i = 0; i < NL; i++) d.l[i] = a.l[i] + Fr::KP[K][i] - b.l[i];
            fe_reduce_small_2p(d);
        } else {
            // every lane multiplies (outside the by-class rounds w^0 = 1 sits next to other twiddles in a wave), so the
            // difference can go into the product un-normalised
            Fe<Fr> w, raw;
            fe_sub_raw<Fr, BC>(raw, a, b);
            pq.load(w, di << RND);
            fe_mul(d, raw, w);
        }
        u.store(s, base + i0);
        u.store(d, base + i1);
        __syncthreads();
        Rounds<Fr, DEG, RND + 1, NEXT>::run(u, pq, blk_base, t, tid, full_tile);
    }
};
template <class Fr, int DEG, int BC>
struct Rounds<Fr, DEG, DEG, BC> {
    static constexpr int FINAL = BC;
    __device__ __forceinline__ static void run(const TilePlanes &, const TwiddlePlanes &, unsigned, unsigned, unsigned, bool) {}
};

struct PassArgs {
    const u32 *x;
    u32 *y;
    const u32 *pq;  // 2^(DEG-1) butterfly twiddles
    const u32 *ta;  // lgp + deg <= 16: (w^e)^t, t < 2^(lgp+deg).  Wider: row k_lo holds (w^e)^(k_lo i), k_lo < 2^split
    const u32 *tb;  // wider only: row k_hi holds (w^(e 2^split))^(k_hi i)
    unsigned log_n;
    unsigned lgp;
    unsigned split;       // 0: one lookup at k*i.  Else k = k_hi 2^split + k_lo and the twiddle is ta[k_lo][i] * tb[k_hi][i] (split = 16 - deg)
    unsigned force_tw;    // multiply by ta[0] even when m == 0 (ta carries the n^-1 factor)
    unsigned tile_elems;  // min(TILE, n)
    unsigned strided_out; // write output i of sub-transform blk to blk + i*S (the input's own layout)
    unsigned canonical;   // reduce the outputs to [0, p): the last pass of a transform; earlier passes stop at [0, 2p)
    unsigned br_in;       // first pass only: the caller's input is in bit-reversed order (element j sits at bitrev(j))
    unsigned br_out;      // last pass only: leave the output in bit-reversed order (y[k] goes to bitrev(k))
};

template <class Fr, int DEG>
__global__ void __launch_bounds__(512) k_ntt_pass(PassArgs A)
{
    constexpr unsigned R = 1u << DEG;
    __shared__ u32 s_u[NL * TilePlanes::STRIDE];
    __shared__ u32 s_pq[NL * TwiddlePlanes::STRIDE];
    const TilePlanes u{s_u};
    const TwiddlePlanes pq{s_pq};
    const unsigned tid = threadIdx.x;
    const unsigned TE = A.tile_elems;
    const unsigned B = TE >> DEG;              // sub-transforms in this tile
    const unsigned S = (1u << A.log_n) >> DEG; // stride between the inputs of one sub-transform
    const unsigned p = 1u << A.lgp;
    const unsigned blk0 = blockIdx.x * B;

    if (tid < (R >> 1)) {
        Fe<Fr> w;
        load_tw(w, A.pq, tid);
        pq.store(w, tid);
    }
    for (unsigned e = tid; e < TE; e += 512) {
        unsigned b, i;
        size_t src_index;
        if (A.br_in) {
            // element j = blk + i*S of the natural order sits at bitrev(j) = (bitrev(blk) << DEG) + bitrev(i): the 2^DEG inputs of
            // a sub-transform are one contiguous run, read in storage order and dropped into LDS at their natural index
            b = e / R;
            const unsigned r = e % R;
            i = bitrev(r, DEG);
            src_index = ((size_t)bitrev0(blk0 + b, A.log_n - DEG) << DEG) + r;
        } else {
            b = e % B;
            i = e / B;
            src_index = (size_t)(blk0 + b) + (size_t)i * S;
        }
        const unsigned blk = blk0 + b;
        Fe<Fr> v;
        load_elem(v, A.x + src_index * 8);
        if (A.lgp != 0 || A.force_tw) {
            const unsigned k = blk & (p - 1);
            if (k * i != 0 || A.force_tw) {
                Fe<Fr> tw;
                if (A.split == 0)
                    load_tw(tw, A.ta, k * i);
                else {
                    // exponent k*i of up to 28 bits: two lookups and a product.  Both tables are laid out [k part][i], so the four
                    // adjacent sub-transforms of a tile read four adjacent rows of ta and ONE row of tb -- contiguous and shared with
                    // the neighbouring tiles -- instead of 48-byte entries scattered over a 3 MB table by k*i mod 2^16 (2.2 x the
                    // pass's input in L2 misses, profiles/r02_fetch_size_calibration.txt)
                    const unsigned k_lo = k & ((1u << A.split) - 1), k_hi = k >> A.split;
                    load_tw(tw, A.ta, (k_lo << DEG) | i);
                    if (k_hi != 0) {
                        Fe<Fr> t2;
                        load_tw(t2, A.tb, (k_hi << DEG) | i);
                        fe_mul(tw, tw, t2);
                    }
                }
                fe_mul(v, v, tw);
            }
        }
        u.store(v, b * R + i);
    }
    __syncthreads();

    // butterflies: thread -> (sub-transform, butterfly index)
    const unsigned half = R >> 1;
    const unsigned sub = tid / half, t = tid % half;
    const bool active = tid < (TE >> 1);
    // every thread passes the DEG barriers of the rounds; threads beyond a short tile only wait
    if (active)
        Rounds<Fr, DEG, 0, 2>::run(u, pq, sub * R, t, tid, TE == TILE); // inputs below 2p: the caller's canonical elements, this kernel's own inter-pass outputs, or a fe_mul result
    else
        for (int r = 0; r < DEG; r++) __syncthreads();

    constexpr int FB = Rounds<Fr, DEG, 0, 2>::FINAL;
    static_assert(FB < 512, "final bound must fit fe_reduce_small (values below 2^9 p)");
    for (unsigned e = tid; e < TE; e += 512) {
        unsigned b, i, lds_i;
        size_t dst_index;
        if (A.br_out) {
            // last pass (lgp + DEG == log_n, or a single pass): y[blk + i p] goes to bitrev(blk + i p) = (bitrev(blk) << DEG) + bitrev(i);
            // LDS holds the outputs in bit-reversed order already, so a sub-transform leaves as one contiguous run, copied straight out
            b = e / R;
            lds_i = e % R;
            i = lds_i;
            dst_index = ((size_t)bitrev0(blk0 + b, A.lgp) << DEG) + lds_i;
        } else {
            if (A.strided_out) {
                b = e % B;
                i = e / B;
                dst_index = (size_t)(blk0 + b) + (size_t)i * S;
            } else if (A.lgp == 0) {
                b = e / R;
                i = e % R;
                dst_index = ((size_t)(blk0 + b) << DEG) + i;
            } else {
                b = e % B;
                i = e / B;
                const unsigned blk = blk0 + b, k = blk & (p - 1);
                dst_index = ((size_t)(blk - k) << DEG) + k + (size_t)i * p;
            }
            lds_i = bitrev(i, DEG);
        }
        Fe<Fr> v;
        u.load(v, b * R + lds_i);
        fe_reduce_small_2p(v);
        if (A.canonical) fe_reduce_once(v); // between passes < 2p is enough: it fits the 32 bytes and the next pass's bounds
        store_elem(A.y + dst_index * 8, v);
    }
}

// x[i] *= ta[i & 0xffff] * tb[i >> 16]   (the inter-slab twiddle w^(r k2) of the multi-GPU transform)
template <class Fr>
__global__ void __launch_bounds__(256) k_slab_twiddle(u32 *__restrict__ x, const u32 *__restrict__ ta, const u32 *__restrict__ tb, unsigned count)
{
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fe<Fr> v, tw;
    load_elem(v, x + (size_t)i * 8);
    load_tw(tw, ta, i & 0xffffu);
    if (i >> 16) {
        Fe<Fr> t2;
        load_tw(t2, tb, i >> 16);
        fe_mul(tw, tw, t2);
    }
    fe_mul(v, v, tw);
    fe_reduce_once(v);
    store_elem(x + (size_t)i * 8, v);
}

// ------------------------------------------------------------------------------- host side

std::mutex g_omega_mutex;
u32 g_omega_wire[8];
bool g_omega_set = false;

// sized for the (w, floor(w R / p)) pairs of k_ntt_pass8; k_ntt_pass's 48-byte entries fit too
const size_t SZ_TA = panda::align256((size_t)(1u << 16) * panda_ntt8::TW2_STRIDE * 4);
const size_t SZ_TB = panda::align256((size_t)(1u << 16) * panda_ntt8::TW2_STRIDE * 4);
const size_t SZ_PQ = panda::align256((size_t)128 * panda_ntt8::TW2_STRIDE * 4);
// plans with a radix-512 pass (k_ntt_pass9): inter-pass tables of up to 2^18 entries, 256 butterfly twiddles
const size_t SZ_T18 = panda::align256((size_t)(1u << 18) * panda_ntt8::TW2_STRIDE * 4);
const size_t SZ_PQ9 = panda::align256((size_t)256 * panda_ntt8::TW2_STRIDE * 4);

// The radices of a transform's passes.  The reference's loop (fft.cu:171-216) takes eight bits per pass and whatever is left in the
// last one; sizes whose bit count leaves a short last pass behind three (two) full ones run one pass less with radix-512 passes in
// front: 2^17 = 9 + 8, 2^18 = 9 + 9, 2^25 = 9 + 8 + 8, 2^26 = 9 + 9 + 8, 2^27 = 9 + 9 + 9 (bit-reversed orderings: see plan_passes).
// *flag = passes & 1 either way: callers read it (unit.rs:458-470).
struct PassPlan {
    unsigned count = 0;
    unsigned d[8] = {0};
    bool wide = false; // contains a radix-512 pass
};

PassPlan plan_passes(unsigned log_n, bool br_in = false, bool br_out = false)
{
    PassPlan pl;
    // radix-512 passes in front; the address maps of the bit-reversed orderings live in the radix-256 kernel, so a bit-reversed INPUT wants
    // that kernel first (the radix-512 passes then go last: 8 + 9, 8 + 8 + 9, 8 + 9 + 9) and a bit-reversed OUTPUT wants it last, which
    // the plans with a radix-256 tail already give; 2^18 / 2^27 (all passes radix 512) and both orderings at once keep the eight-bit plan
    unsigned nines = 0;
    switch (log_n) {
    case 17: case 25: nines = 1; break;
    case 18: case 26: nines = 2; break;
    case 27: nines = 3; break;
    default: break;
    }
    if ((br_in && br_out) || ((br_in || br_out) && nines * 9 == log_n)) nines = 0;
    pl.wide = nines != 0;
    unsigned eights[8], ne = 0, left = log_n - 9 * nines;
    while (left) {
        const unsigned deg = left < 8 ? left : 8;
        eights[ne++] = deg;
        left -= deg;
    }
    if (br_in) { // radix-256 passes first (with nines != 0 they are all full: 17, 25, 26 leave multiples of eight)
        for (unsigned j = 0; j < ne; j++) pl.d[pl.count++] = eights[j];
        for (unsigned j = 0; j < nines; j++) pl.d[pl.count++] = 9;
    } else {
        for (unsigned j = 0; j < nines; j++) pl.d[pl.count++] = 9;
        for (unsigned j = 0; j < ne; j++) pl.d[pl.count++] = eights[j];
    }
    return pl;
}

// Streamed inter-pass table (panda_ntt_set_streamed_tables): where a radix-256 pass would multiply its outputs by TWO table entries
// (the twiddle's index range, 2^(log_p + 8 + deg2), exceeds the 2^16-entry tables: the second boundary of every three-pass transform),
// it can read ONE 32-byte entry per element from a table over the whole range instead -- as large as the data, streamed beside it,
// built once per root and size and cached -- and multiply once.
// Modes (panda_ntt_set_streamed_tables): 0 = off; 1 = the built-in policy: tables of up to 2^24 entries (512 MiB per direction and host
// thread; -6 % at 2^20, -4 % at 2^22, -1.4 % at 2^24, profiles/r05_ntt_streamed_table.txt); 2 = also the 1 GiB / 2 GiB tables of 2^25 /
// 2^26 points (-1.7 % / -0.9 %: not worth their memory by default); 3 = test hook: mode 1 with the table's allocation treated as failed
// (the out-of-memory fallback below).
constexpr unsigned STREAMED_POLICY = 1, STREAMED_LARGE = 2, STREAMED_FAIL_ALLOC = 3;
constexpr unsigned STREAMED_POLICY_MAX_BITS = 24;
std::atomic<unsigned> g_streamed_tables{STREAMED_POLICY};

// log2 of the entries of the streamed table of pass j, 0 if that pass does not take one
unsigned streamed_table_bits(const PassPlan &pl, unsigned j, unsigned log_n, unsigned mode)
{
    if (!mode || log_n < 11 || j + 1 >= pl.count || (pl.d[j] != 8 && pl.d[j] != 9)) return 0;
    unsigned log_p = 0;
    for (unsigned i = 0; i < j; i++) log_p += pl.d[i];
    const unsigned bits = log_p + pl.d[j] + pl.d[j + 1], cap = pl.wide ? 18u : 16u; // up to 2^cap entries the pass reads one table anyway
    const unsigned most = mode == STREAMED_LARGE ? (unsigned)POW_BITS : STREAMED_POLICY_MAX_BITS;
    return (bits > cap && bits <= most) ? bits : 0;
}

// A streamed table that did not fit once is not tried again by this host thread on that device (until panda_ntt_set_streamed_tables or
// panda_ntt_tear_down): the transform runs with the two small tables under the key WITHOUT the streamed bit, so later calls hit the cache
// instead of failing the large hipMalloc, dropping the other cache slot and rebuilding every table each time.
struct StreamedUnavailable {
    int device = -1;
    unsigned bits = 0xffffffffu; // tables of at least 2^bits entries
};
thread_local StreamedUnavailable g_streamed_unavailable;
thread_local uint64_t g_table_builds = 0; // whole-transform table sets built by this host thread (panda_ntt_table_builds)

struct PassTables {
    size_t ta, tb, pq;
};
PassTables pass_tables(const PassPlan &pl, unsigned j, unsigned log_n = 0, unsigned streamed = 0)
{
    if (const unsigned bits = streamed_table_bits(pl, j, log_n, streamed)) return PassTables{panda::align256((size_t)32 << bits), 256, pl.wide ? SZ_PQ9 : SZ_PQ};
    if (!pl.wide) return PassTables{SZ_TA, SZ_TB, SZ_PQ};
    const bool last = j + 1 == pl.count;
    return PassTables{last ? 256 : SZ_T18, last ? 256 : SZ_T18, SZ_PQ9};
}
// bytes ntt_passes carves out of its arena for a transform of 2^log_n points
size_t passes_table_bytes(unsigned log_n, bool br_in = false, bool br_out = false, unsigned streamed = 0)
{
    const PassPlan pl = plan_passes(log_n, br_in, br_out);
    size_t total = 0;
    for (unsigned j = 0; j < pl.count; j++) {
        const PassTables t = pass_tables(pl, j, log_n, streamed);
        total += t.ta + t.tb + t.pq + 3 * 256;
    }
    return total;
}

template <class Fr>
void fill_pow_base(PowBase &pb, const Fe<Fr> &base, const Fe<Fr> *scale)
{
    Fe<Fr> cur = base;
    for (int j = 0; j < POW_BITS; j++) {
        Fe<Fr> c = cur;
        fe_reduce_once(c);
        for (int i = 0; i < NL; i++) pb.pw[j][i] = c.l[i];
        fe_sqr(cur, cur);
    }
    pb.has_scale = scale ? 1 : 0;
    for (int i = 0; i < NL; i++) pb.scale[i] = scale ? scale->l[i] : 0;
}

template <class Fr>
void build_table(hipStream_t stream, const Fe<Fr> &base, const Fe<Fr> *scale, unsigned count, u32 *d_out, unsigned rows_deg = 0)
{
    PowBase pb;
    fill_pow_base<Fr>(pb, base, scale);
    hipLaunchKernelGGL(k_pow_table<Fr>, dim3((count + 255) / 256), dim3(256), 0, stream, pb, count, rows_deg, d_out);
}

template <class Fr>
void build_table2(hipStream_t stream, const Fe<Fr> &base, const Fe<Fr> *scale, unsigned count, u32 *d_out, unsigned rows_deg = 0)
{
    PowBase pb;
    fill_pow_base<Fr>(pb, base, scale);
    hipLaunchKernelGGL(k_pow_table2<Fr>, dim3((count + 255) / 256), dim3(256), 0, stream, pb, count, rows_deg, d_out);
}

// LDS planes per exchange batch and resident workgroups per CU of k_ntt_pass8 (5 planes: 40 KB + 10 KB of twiddles, three fit)
constexpr int P8_PLANES = 5, P8_MINW = 3;

template <class Fr>
void launch_pass8(bool first, bool last, const panda_ntt8::Pass8Args &a, unsigned tiles, hipStream_t s)
{
    using namespace panda_ntt8;
    if (last)
        hipLaunchKernelGGL((k_ntt_pass8<Fr, false, true, P8_PLANES, P8_MINW>), dim3(tiles), dim3(THREADS), 0, s, a);
    else if (first)
        hipLaunchKernelGGL((k_ntt_pass8<Fr, true, false, P8_PLANES, P8_MINW>), dim3(tiles), dim3(THREADS), 0, s, a);
    else
        hipLaunchKernelGGL((k_ntt_pass8<Fr, false, false, P8_PLANES, P8_MINW>), dim3(tiles), dim3(THREADS), 0, s, a);
}

// radix-512 pass: exchange batches of three limb planes (24 KB + 20 KB of twiddles), three workgroups per CU
constexpr int P9_PLANES = 3, P9_MINW = 3;

template <class Fr>
void launch_pass9(bool first, bool last, const panda_ntt8::Pass8Args &a, unsigned tiles, hipStream_t s)
{
    using namespace panda_ntt8;
    if (last)
        hipLaunchKernelGGL((k_ntt_pass9<Fr, false, true, P9_PLANES, P9_MINW>), dim3(tiles), dim3(THREADS), 0, s, a);
    else if (first)
        hipLaunchKernelGGL((k_ntt_pass9<Fr, true, false, P9_PLANES, P9_MINW>), dim3(tiles), dim3(THREADS), 0, s, a);
    else
        hipLaunchKernelGGL((k_ntt_pass9<Fr, false, false, P9_PLANES, P9_MINW>), dim3(tiles), dim3(THREADS), 0, s, a);
}

// short last pass / size-G slab transforms in registers (radix 2, 4, 8): d_pq holds max(1, 2^(deg-1)) precomputed-quotient entries
template <class Fr>
void launch_small(unsigned deg, const u32 *x, u32 *y, const u32 *d_pq, unsigned log_count, bool br_out, hipStream_t s)
{
    using namespace panda_ntt8;
    const SmallArgs a{x, y, d_pq, log_count, br_out ? 1u : 0u};
    const unsigned blocks = (unsigned)((((u64)1 << log_count) + 255) / 256);
    if (deg == 1)
        hipLaunchKernelGGL((k_ntt_small<Fr, 1>), dim3(blocks), dim3(256), 0, s, a);
    else if (deg == 2)
        hipLaunchKernelGGL((k_ntt_small<Fr, 2>), dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_ntt_small<Fr, 3>), dim3(blocks), dim3(256), 0, s, a);
}

template <class Fr>
void launch_pass(unsigned deg, const PassArgs &a, unsigned tiles, hipStream_t s)
{
    switch (deg) {
    case 1: hipLaunchKernelGGL((k_ntt_pass<Fr, 1>), dim3(tiles), dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_ntt_pass<Fr, 2>), dim3(tiles), dim3(512), 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_ntt_pass<Fr, 3>), dim3(tiles), dim3(512), 0, s, a); break;
    case 4: hipLaunchKernelGGL((k_ntt_pass<Fr, 4>), dim3(tiles), dim3(512), 0, s, a); break;
    case 5: hipLaunchKernelGGL((k_ntt_pass<Fr, 5>), dim3(tiles), dim3(512), 0, s, a); break;
    case 6: hipLaunchKernelGGL((k_ntt_pass<Fr, 6>), dim3(tiles), dim3(512), 0, s, a); break;
    case 7: hipLaunchKernelGGL((k_ntt_pass<Fr, 7>), dim3(tiles), dim3(512), 0, s, a); break;
    default: hipLaunchKernelGGL((k_ntt_pass<Fr, 8>), dim3(tiles), dim3(512), 0, s, a); break;
    }
}

// The caller's earlier work on other (blocking) streams must be visible: the reference runs its passes on the
// legacy NULL stream (fft.cu:201), which implies exactly this dependency.
struct OrderEvent { // one per host thread and device, created on first use: a transform is 2 ms, an event create + destroy per call showed
    hipEvent_t ev = nullptr;
    int device = -1;
    ~OrderEvent()
    {
        if (ev) (void)hipEventDestroy(ev);
    }
};
thread_local OrderEvent g_order_event;

hipError_t order_after_null_stream(hipStream_t stream)
{
    if (stream == nullptr) return hipSuccess; // the NULL stream is ordered after itself
    int dev = -1;
    PANDA_TRY(hipGetDevice(&dev));
    OrderEvent &oe = g_order_event;
    if (oe.ev && oe.device != dev) {
        (void)hipEventDestroy(oe.ev);
        oe.ev = nullptr;
    }
    if (!oe.ev) {
        PANDA_TRY(hipEventCreateWithFlags(&oe.ev, hipEventDisableTiming));
        oe.device = dev;
    }
    PANDA_TRY(hipEventRecord(oe.ev, nullptr));
    PANDA_TRY(hipStreamWaitEvent(stream, oe.ev, 0));
    return hipSuccess;
}

// Twiddle tables of the last transform run by this host thread, kept on the device: a prover transforms many
// polynomials with the same root, and rebuilding three small tables per pass costs ~0.1 ms of launches per call.
struct TwiddleCache {
    void *base = nullptr;
    size_t capacity = 0, used = 0;
    int device = -1;
    bool valid = false;
    u32 key[12] = {0};
    // tables enqueued by a call that did not wait for its stream (the *_enqueue entry points) are complete only in that
    // stream's order.  The enqueueing call records an event behind them; the next call makes ITS stream wait for that event --
    // no host wait, and no handle of the earlier stream is kept (the caller may have destroyed it since).
    hipEvent_t pending_ev = nullptr;
    int pending_dev = -1;
    bool has_pending = false;
    hipError_t mark_pending(hipStream_t s)
    {
        int dev = -1;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (pending_ev && pending_dev != dev) {
            (void)hipEventDestroy(pending_ev);
            pending_ev = nullptr;
        }
        if (!pending_ev) {
            if ((e = hipEventCreateWithFlags(&pending_ev, hipEventDisableTiming)) != hipSuccess) return e;
            pending_dev = dev;
        }
        if ((e = hipEventRecord(pending_ev, s)) != hipSuccess) return e;
        has_pending = true;
        return hipSuccess;
    }
    hipError_t settle(hipStream_t next)
    {
        if (!has_pending) return hipSuccess;
        has_pending = false; // whatever happens below, the next call starts clean
        return hipStreamWaitEvent(next, pending_ev, 0);
    }
    void drop_pending()
    {
        has_pending = false;
        if (pending_ev) (void)hipEventDestroy(pending_ev);
        pending_ev = nullptr;
    }
    hipError_t ensure(size_t bytes)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (base && (dev != device || bytes > capacity)) {
            (void)hipDeviceSynchronize();
            (void)hipFree(base);
            base = nullptr;
            valid = false;
        }
        if (!base) {
            e = hipMalloc(&base, bytes);
            if (e != hipSuccess) return e;
            capacity = bytes;
            device = dev;
            valid = false;
        }
        used = 0;
        return hipSuccess;
    }
    void *take(size_t bytes)
    {
        size_t off = (used + 255) & ~(size_t)255;
        if (off + bytes > capacity) return nullptr;
        used = off + bytes;
        return (char *)base + off;
    }
    TwiddleCache() = default;
    TwiddleCache(const TwiddleCache &) = delete;
    TwiddleCache &operator=(const TwiddleCache &) = delete;
    ~TwiddleCache()
    {
        drop_pending();
        if (base) (void)hipFree(base); // a host thread that exits without panda_ntt_tear_down() does not leak its tables
    }
    hipError_t release()
    {
        hipError_t e = hipSuccess;
        if (base) {
            (void)hipDeviceSynchronize();
            e = hipFree(base);
        }
        base = nullptr;
        capacity = used = 0;
        valid = false;
        drop_pending();
        return e;
    }
};
// one entry per call family, so that the alternating steps of a sharded transform do not evict each other's tables
// (whole transforms get two entries: a prover alternates between a size's forward and inverse transform, and the streamed inter-pass
// table of a 2^24-point transform is 2^24 twelve-product entries to rebuild)
enum { TW_WHOLE = 0, TW_WHOLE_B = 1, TW_SLAB1 = 2, TW_SLAB2 = 3, TW_SLOTS = 4 };
thread_local TwiddleCache g_twiddles[TW_SLOTS];
thread_local unsigned g_whole_last = 0; // which of the two whole-transform entries was used last

template <class Fr>
void twiddle_key(u32 (&key)[12], unsigned log_n, unsigned variant, const u32 *omega_wire)
{
    key[0] = Fr::PW[0] ^ Fr::PW[7];
    key[1] = log_n;
    key[2] = variant;
    for (int i = 0; i < 8; i++) key[3 + i] = omega_wire[i];
    key[11] = 0;
}

// Device time of the last whole transform on this host thread: HIP events on the launch stream around the passes (what the MSM's
// phase timers are for its kernels), read by panda_ntt_last_device_ms.  A synchronous call timed from outside also counts the host's
// wake-up after the last pass (~0.1-0.2 ms of a 2 ms transform).
struct PassTimer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int device = -1;
    float ms = 0.f;
    hipError_t begin(hipStream_t s)
    {
        int dev = -1;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (e0 && device != dev) drop();
        if (!e0) {
            if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
            if ((e = hipEventCreate(&e1)) != hipSuccess) return e;
            device = dev;
        }
        return hipEventRecord(e0, s);
    }
    hipError_t end(hipStream_t s) { return hipEventRecord(e1, s); }
    void read()
    {
        float v = 0.f;
        if (hipEventElapsedTime(&v, e0, e1) == hipSuccess) ms = v;
    }
    void drop()
    {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        e0 = e1 = nullptr;
    }
    ~PassTimer() { drop(); }
};
thread_local PassTimer g_pass_timer;

// All passes of one local transform of size 2^log_n with root `omega` (internal form).  `scale`, when given,
// multiplies every output (folded into the last pass's twiddles).  Leaves the result in src when *passes_out
// is even, in dst when odd; enqueues only.
// `build` = false reuses the tables already sitting in `arena` (same carve order).
template <class Fr, class Alloc>
hipError_t ntt_passes(hipStream_t stream, Alloc &arena, const u32 *src, u32 *dst, const Fe<Fr> &omega, unsigned log_n, const Fe<Fr> *scale,
                      unsigned *passes_out, bool build = true, bool br_in = false, bool br_out = false, unsigned streamed = 0)
{
    const u64 n = (u64)1 << log_n;
    unsigned log_p = 0, passes = 0;
    const PassPlan pl = plan_passes(log_n, br_in, br_out);
    const unsigned total_passes = pl.count;
    // Transforms of 2^11 points and more run their radix-256 / radix-512 passes in k_ntt_pass8 / k_ntt_pass9, which multiply the twiddle
    // between two passes onto the OUTPUT of the earlier one; a shorter last pass (k_ntt_small) then finds it done.  Smaller transforms
    // keep the twiddle on the input side of k_ntt_pass.
    const bool regs8 = log_n >= 11;
    const unsigned cap = pl.wide ? 18 : 16; // log2 of the largest inter-pass table
    while (log_p < log_n) {
        const unsigned deg = pl.d[passes];
        const bool last = (passes + 1 == total_passes);
        const PassTables sz = pass_tables(pl, passes, log_n, streamed);
        const unsigned stream_bits = streamed_table_bits(pl, passes, log_n, streamed);
        u32 *d_ta = (u32 *)arena.take(sz.ta), *d_tb = (u32 *)arena.take(sz.tb), *d_pq = (u32 *)arena.take(sz.pq);
        if (!d_ta || !d_tb || !d_pq) return hipErrorOutOfMemory;
        Fe<Fr> base;
        if (regs8 && (deg == 9 || deg == 8 || (last && deg >= 4))) { // a last pass of radix 16 ... 128 runs as k_ntt_pass8 with its first 8 - deg rounds off
            panda_ntt8::Pass8Args a{};
            const unsigned full = deg == 9 ? 9 : 8; // rounds of the kernel that runs this pass
            a.skip = full - deg;
            a.x = src;
            a.y = dst;
            a.pq = d_pq;
            a.ta = d_ta;
            a.tb = d_tb;
            a.log_n = log_n;
            a.lgp = log_p;
            a.br_in = (br_in && passes == 0) ? 1 : 0;
            a.br_out = (br_out && last) ? 1 : 0;
            a.tiles = (unsigned)(n / panda_ntt8::ELEMS);
            const unsigned deg2 = last ? 0 : pl.d[passes + 1]; // radix of the next pass (deg == full unless last)
            if (!last) {
                // twiddle W^(i2 k2), W = w^(n / 2^(log_p + deg + deg2)), k2 < 2^(log_p + deg), i2 < 2^deg2; tables of at most 2^cap entries
                if (log_p + deg + deg2 <= cap) {
                    a.ca = log_p + deg;
                    a.cb = 0;
                } else {
                    a.ca = std::min(log_p, cap - deg2);
                    a.cb = log_p + deg - a.ca;
                }
                a.i2_shift = log_n - deg2 - log_p - deg;
                a.wide = stream_bits ? 1u : 0u;
            }
            if (build) {
                fe_pow_u64(base, omega, n >> full); // butterfly twiddles (w^(n / 2^full))^t
                build_table2<Fr>(stream, base, nullptr, 1u << (full - 1), d_pq);
                if (!last && stream_bits) { // [i2][k2], k2 < 2^(log_p + 8): exponent i2 k2 < 2^stream_bits
                    fe_pow_u64(base, omega, n >> (log_p + deg) >> deg2);
                    PowBase pb;
                    fill_pow_base<Fr>(pb, base, nullptr);
                    hipLaunchKernelGGL(k_pow_table_wire<Fr>, dim3((unsigned)((((u64)1 << stream_bits) + 255) / 256)), dim3(256), 0, stream, pb, 1u << stream_bits, log_p + deg, d_ta);
                } else if (!last) {
                    fe_pow_u64(base, omega, n >> (log_p + deg) >> deg2);
                    build_table2<Fr>(stream, base, (scale && passes == 0) ? scale : nullptr, 1u << (deg2 + a.ca), d_ta, a.ca);
                    if (a.cb) {
                        Fe<Fr> base_b;
                        fe_pow_u64(base_b, base, (u64)1 << a.ca);
                        build_table2<Fr>(stream, base_b, nullptr, 1u << (deg2 + a.cb), d_tb, a.cb);
                    }
                }
            }
            if (deg == 9)
                launch_pass9<Fr>(passes == 0, last, a, (unsigned)(n / panda_ntt8::ELEMS), stream);
            else
                launch_pass8<Fr>(passes == 0, last, a, (unsigned)(n / panda_ntt8::ELEMS), stream);
        } else if (regs8 && last && deg <= 3) {
            // a short last pass behind k_ntt_pass8 (its twiddle is already on the data): radix 2 / 4 / 8 in registers
            if (build) {
                fe_pow_u64(base, omega, n >> deg);
                build_table2<Fr>(stream, base, nullptr, std::max(1u, (1u << deg) >> 1), d_pq);
            }
            launch_small<Fr>(deg, src, dst, d_pq, log_n - deg, br_out, stream);
        } else {
            PassArgs a{};
            a.x = src;
            a.y = dst;
            a.pq = d_pq;
            a.ta = d_ta;
            a.tb = d_tb;
            a.log_n = log_n;
            a.lgp = log_p;
            a.tile_elems = (unsigned)std::min<u64>(TILE, n);
            const unsigned mbits = log_p + deg; // bits of the twiddle exponent k * i
            // (this kernel only runs transforms below 2^11 points: from there on every pass is k_ntt_pass8 / k_ntt_small, see above)
            a.split = (log_p != 0 && mbits > 16) ? 16 - deg : 0;
            a.force_tw = (scale && last) ? 1 : 0;
            a.strided_out = 0;
            a.canonical = last ? 1 : 0;
            a.br_in = (br_in && passes == 0) ? 1 : 0;
            a.br_out = (br_out && last) ? 1 : 0;
            if (build) {
                fe_pow_u64(base, omega, n >> deg); // butterfly twiddles (w^(n >> deg))^t
                build_table<Fr>(stream, base, nullptr, std::max(1u, (1u << deg) >> 1), d_pq);
                if (log_p != 0) {
                    fe_pow_u64(base, omega, n >> log_p >> deg);
                    if (a.split == 0)
                        build_table<Fr>(stream, base, a.force_tw ? scale : nullptr, 1u << mbits, d_ta);
                    else {
                        build_table<Fr>(stream, base, a.force_tw ? scale : nullptr, 1u << 16, d_ta, deg); // [k_lo][i], k_lo < 2^split
                        Fe<Fr> base_b;
                        fe_pow_u64(base_b, base, (u64)1 << a.split);
                        build_table<Fr>(stream, base_b, nullptr, 1u << (log_p - a.split + deg), d_tb, deg); // [k_hi][i], k_hi < 2^(log_p - split)
                    }
                } else if (a.force_tw) {
                    fe_one(base);
                    build_table<Fr>(stream, base, scale, 1, d_ta); // single pass: the table is just the scale
                }
            }
            launch_pass<Fr>(deg, a, (unsigned)(n / a.tile_elems), stream);
        }
        PANDA_TRY(hipGetLastError());
        const u32 *tmp = dst;
        dst = const_cast<u32 *>(src);
        src = tmp;
        log_p += deg;
        passes++;
    }
    *passes_out = passes;
    return hipSuccess;
}

template <class Fr>
void inverse_parameters(Fe<Fr> &omega, Fe<Fr> &scale, u64 n)
{
    Fe<Fr> oi, nfe;
    fe_inv(oi, omega);
    omega = oi;
    fe_from_u32(nfe, (u32)n); // n <= 2^28
    fe_inv(scale, nfe);
    fe_reduce_once(scale);
}

template <class Fr>
hipError_t ntt_run(hipStream_t stream, void *d_src, void *d_dst, const u32 *omega_wire, unsigned log_n, unsigned *flag, bool inverse, bool br_in = false,
                   bool br_out = false)
{
    if (log_n > 28 || !d_src || !d_dst || !omega_wire) return hipErrorInvalidValue;
    if (panda::extent_too_short(d_src, (size_t)32 << log_n) || panda::extent_too_short(d_dst, (size_t)32 << log_n)) return hipErrorInvalidValue;
    PANDA_TRY(order_after_null_stream(stream));
    u32 key[12];
    int dev = -1;
    PANDA_TRY(hipGetDevice(&dev));
    const unsigned mode = g_streamed_tables.load(std::memory_order_relaxed);
    unsigned streamed = mode == STREAMED_FAIL_ALLOC ? STREAMED_POLICY : mode, streamed_bits = 0;
    {
        const PassPlan pl = plan_passes(log_n, br_in, br_out);
        for (unsigned j = 0; j < pl.count; j++) streamed_bits = std::max(streamed_bits, streamed_table_bits(pl, j, log_n, streamed));
    }
    if (!streamed_bits || (g_streamed_unavailable.device == dev && streamed_bits >= g_streamed_unavailable.bits)) streamed = 0;
    const unsigned variant = (inverse ? 1u : 0u) | (br_in ? 2u : 0u) | (br_out ? 4u : 0u); // the bit-reversed orderings may run another plan
    twiddle_key<Fr>(key, log_n, variant | (streamed << 3), omega_wire);
    // the entry that holds these tables, else the one that was not used last
    unsigned slot = g_whole_last ^ 1u;
    for (unsigned c = 0; c < 2; c++) {
        const TwiddleCache &t = g_twiddles[TW_WHOLE + c];
        if (t.valid && t.device == dev && memcmp(key, t.key, sizeof(key)) == 0) slot = c;
    }
    g_whole_last = slot;
    TwiddleCache &tw = g_twiddles[TW_WHOLE + slot];
    PANDA_TRY(tw.settle(stream));
    const bool hit = tw.valid && tw.device == dev && memcmp(key, tw.key, sizeof(key)) == 0;
    Fe<Fr> omega, scale;
    fe_zero(omega);
    fe_zero(scale);
    if (!hit) { // host-side parameters (two Fermat inversions for the inverse transform) only when tables are rebuilt
        fe_from_wire(omega, omega_wire);
        if (inverse) inverse_parameters<Fr>(omega, scale, (u64)1 << log_n);
        hipError_t got = (streamed && mode == STREAMED_FAIL_ALLOC) ? hipErrorOutOfMemory : tw.ensure(passes_table_bytes(log_n, br_in, br_out, streamed) + 4096);
        if (got == hipErrorOutOfMemory && streamed) {
            // no room for a table as large as the data: the transform runs with the two small tables, as it did before round 5 -- and so do
            // this thread's later transforms of this size and larger on this device (g_streamed_unavailable)
            (void)hipGetLastError();
            if (g_streamed_unavailable.device != dev) g_streamed_unavailable = StreamedUnavailable{dev, streamed_bits};
            g_streamed_unavailable.bits = std::min(g_streamed_unavailable.bits, streamed_bits);
            streamed = 0;
            twiddle_key<Fr>(key, log_n, variant, omega_wire);
            got = tw.ensure(passes_table_bytes(log_n, br_in, br_out, 0) + 4096);
        }
        PANDA_TRY(got);
        g_table_builds++;
    } else
        tw.used = 0;
    tw.valid = false;
    unsigned passes = 0;
    PassTimer &pt = g_pass_timer;
    // clock stamps around the passes (panda_set_clock_stamps; panda_internal.h): cycles and the clock they ran at, beside the milliseconds
    const bool stamps = panda::clock_stamps_enabled();
    uint64_t *stamp_block = nullptr;
    panda::thread_ntt_clock() = panda::ClockDelta{};
    if (stamps) {
        PANDA_TRY(panda::thread_stamp_blocks(&stamp_block));
        stamp_block += 2 * 2 * panda::CLOCK_STAMP_SLOTS; // blocks 2 and 3 (0 and 1 are the MSM's)
        for (unsigned i = 0; i < 2 * 2 * panda::CLOCK_STAMP_SLOTS; i++) stamp_block[i] = 0;
    }
    PANDA_TRY(pt.begin(stream));
    if (stamps) PANDA_TRY(panda::enqueue_clock_stamp(stream, stamp_block));
    PANDA_TRY(ntt_passes<Fr>(stream, tw, (const u32 *)d_src, (u32 *)d_dst, omega, log_n, inverse ? &scale : nullptr, &passes, !hit, br_in, br_out, streamed));
    if (stamps) PANDA_TRY(panda::enqueue_clock_stamp(stream, stamp_block + 2 * panda::CLOCK_STAMP_SLOTS));
    PANDA_TRY(pt.end(stream));
    if (flag) *flag = passes & 1u;           // fft.cu:211
    PANDA_TRY(hipStreamSynchronize(stream)); // the reference is synchronous on return (fft.cu:202)
    pt.read();
    if (stamps) panda::thread_ntt_clock() = panda::clock_delta(stamp_block, stamp_block + 2 * panda::CLOCK_STAMP_SLOTS);
    memcpy(tw.key, key, sizeof(key));
    tw.valid = true; // only after the tables are known to be complete
    tw.has_pending = false;
    return hipSuccess;
}

// x[j] *= base^j, j < 2^log_n, in place (the coset shift of a coset NTT): two power tables and k_slab_twiddle
template <class Fr>
hipError_t scale_by_powers(hipStream_t stream, u32 *d_x, unsigned log_n, const Fe<Fr> &base)
{
    const u64 n = (u64)1 << log_n;
    panda::Arena &arena = panda::thread_arena();
    PANDA_TRY(arena.reserve(SZ_TA + SZ_TB + 4096));
    u32 *d_ta = (u32 *)arena.take(SZ_TA), *d_tb = (u32 *)arena.take(SZ_TB);
    if (!d_ta || !d_tb) return hipErrorOutOfMemory;
    build_table<Fr>(stream, base, nullptr, (unsigned)std::min<u64>(n, 1u << 16), d_ta);
    if (log_n > 16) {
        Fe<Fr> base_b;
        fe_pow_u64(base_b, base, (u64)1 << 16);
        build_table<Fr>(stream, base_b, nullptr, 1u << (log_n - 16), d_tb);
    }
    hipLaunchKernelGGL(k_slab_twiddle<Fr>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_x, d_ta, d_tb, (unsigned)n);
    PANDA_TRY(hipGetLastError());
    return hipStreamSynchronize(stream);
}

// Coset transforms: forward y[k] = sum_j x[j] g^j w^(jk) (evaluation on the coset g*H), inverse x[j] = g^-j n^-1 sum_k y[k] w^(-jk).
// The shift is applied as one extra element-wise pass before (after) the passes of the plain transform.
template <class Fr>
hipError_t ntt_coset_run(const panda_ntt_configuration_v1 &cfg, const void *shift_wire, bool inverse)
{
    if (!shift_wire || !cfg.d_src || !cfg.d_dst || !cfg.d_omega || !cfg.flag || cfg.log_n > 28) return hipErrorInvalidValue;
    if (panda::extent_too_short(cfg.d_src, (size_t)32 << cfg.log_n) || panda::extent_too_short(cfg.d_dst, (size_t)32 << cfg.log_n)) return hipErrorInvalidValue;
    hipStream_t stream = static_cast<hipStream_t>(cfg.stream.handle);
    Fe<Fr> g;
    fe_from_wire(g, (const u32 *)shift_wire);
    if (fe_is_zero_mod_p(g)) return hipErrorInvalidValue;
    if (!inverse) {
        PANDA_TRY(order_after_null_stream(stream));
        PANDA_TRY(scale_by_powers<Fr>(stream, (u32 *)cfg.d_src, cfg.log_n, g));
        return ntt_run<Fr>(stream, cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, false);
    }
    PANDA_TRY(ntt_run<Fr>(stream, cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, true));
    Fe<Fr> gi;
    fe_inv(gi, g);
    u32 *res = (*(unsigned *)cfg.flag & 1u) ? (u32 *)cfg.d_dst : (u32 *)cfg.d_src;
    return scale_by_powers<Fr>(stream, res, cfg.log_n, gi);
}

// a rank's slab and scratch hold 2^(log_n - log_ranks) elements each (log_n >= log_ranks checked by the callers)
bool slab_too_short(const panda_ntt_slab_configuration &cfg)
{
    const size_t bytes = (size_t)32 << (cfg.log_n - cfg.log_ranks);
    return panda::extent_too_short(cfg.d_slab, bytes) || panda::extent_too_short(cfg.d_scratch, bytes);
}

// shared tail of the two slab steps: publish the tables in the cache, optionally wait
hipError_t slab_finish(TwiddleCache &tw, const u32 (&key)[12], hipStream_t stream, bool wait)
{
    if (wait) {
        PANDA_TRY(hipStreamSynchronize(stream));
        tw.has_pending = false;
    } else
        PANDA_TRY(tw.mark_pending(stream));
    memcpy(tw.key, key, sizeof(tw.key));
    tw.valid = true;
    return hipSuccess;
}

// multi-GPU step 1: local transform of the rank's decimated slab + the inter-slab twiddle w^(rank * k2).
// The twiddle tables are cached per host thread (a prover repeats the same sharded transform), so a repeat call launches only
// the passes and the twiddle sweep and does no host-side field arithmetic.  wait = false enqueues without synchronising.
template <class Fr>
hipError_t slab_step1(const panda_ntt_slab_configuration &cfg, bool wait)
{
    if (cfg.log_ranks > 8 || cfg.log_n > 28 || cfg.log_n < cfg.log_ranks || !cfg.d_slab || !cfg.d_scratch || !cfg.omega) return hipErrorInvalidValue;
    if (cfg.rank >= (1u << cfg.log_ranks)) return hipErrorInvalidValue;
    if (slab_too_short(cfg)) return hipErrorInvalidValue;
    hipStream_t stream = static_cast<hipStream_t>(cfg.stream.handle);
    PANDA_TRY(order_after_null_stream(stream));
    const unsigned log_m = cfg.log_n - cfg.log_ranks;
    const u64 m = (u64)1 << log_m;
    TwiddleCache &tw = g_twiddles[TW_SLAB1];
    u32 key[12];
    twiddle_key<Fr>(key, cfg.log_n, 0x100u | (cfg.log_ranks << 16) | (cfg.rank << 20), (const u32 *)cfg.omega);
    PANDA_TRY(tw.settle(stream));
    int dev = -1;
    PANDA_TRY(hipGetDevice(&dev));
    const bool hit = tw.valid && tw.device == dev && memcmp(key, tw.key, sizeof(key)) == 0;
    Fe<Fr> omega, omega_m;
    fe_zero(omega);
    fe_zero(omega_m);
    if (!hit) {
        fe_from_wire(omega, (const u32 *)cfg.omega);
        fe_pow_u64(omega_m, omega, (u64)1 << cfg.log_ranks); // root of the local size-m transforms
        PANDA_TRY(tw.ensure(passes_table_bytes(log_m) + SZ_TA + SZ_TB + 4096));
    } else
        tw.used = 0;
    tw.valid = false;
    unsigned passes = 0;
    PANDA_TRY(ntt_passes<Fr>(stream, tw, (const u32 *)cfg.d_slab, (u32 *)cfg.d_scratch, omega_m, log_m, nullptr, &passes, !hit));
    u32 *res = (passes & 1u) ? (u32 *)cfg.d_scratch : (u32 *)cfg.d_slab;
    if (cfg.rank != 0) {
        u32 *d_ta = (u32 *)tw.take(SZ_TA), *d_tb = (u32 *)tw.take(SZ_TB);
        if (!d_ta || !d_tb) return hipErrorOutOfMemory;
        if (!hit) {
            Fe<Fr> base, base_b;
            fe_pow_u64(base, omega, cfg.rank);
            build_table<Fr>(stream, base, nullptr, (unsigned)std::min<u64>(m, 1u << 16), d_ta);
            if (log_m > 16) {
                fe_pow_u64(base_b, base, (u64)1 << 16);
                build_table<Fr>(stream, base_b, nullptr, 1u << (log_m - 16), d_tb);
            }
        }
        hipLaunchKernelGGL(k_slab_twiddle<Fr>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, res, d_ta, d_tb, (unsigned)m);
        PANDA_TRY(hipGetLastError());
    }
    if (cfg.flag) *(unsigned *)cfg.flag = passes & 1u; // known without waiting: the parity of the pass count
    return slab_finish(tw, key, stream, wait);
}

// multi-GPU step 2: after the all-to-all the slab holds [j1][k2'] (G x m/G); transforms of size G down j1
template <class Fr>
hipError_t slab_step2(const panda_ntt_slab_configuration &cfg, bool wait, bool inverse = false)
{
    if (cfg.log_ranks > 8 || cfg.log_n > 28 || cfg.log_n < 2 * cfg.log_ranks || !cfg.d_slab || !cfg.d_scratch || !cfg.omega) return hipErrorInvalidValue;
    if (slab_too_short(cfg)) return hipErrorInvalidValue;
    hipStream_t stream = static_cast<hipStream_t>(cfg.stream.handle);
    PANDA_TRY(order_after_null_stream(stream));
    const unsigned log_m = cfg.log_n - cfg.log_ranks;
    const u64 m = (u64)1 << log_m;
    if (cfg.log_ranks == 0) {
        if (cfg.flag) *(unsigned *)cfg.flag = 0;
        if (wait) PANDA_TRY(hipStreamSynchronize(stream));
        return hipSuccess;
    }
    TwiddleCache &tw = g_twiddles[TW_SLAB2];
    u32 key[12];
    twiddle_key<Fr>(key, cfg.log_n, 0x200u | (inverse ? 0x400u : 0u) | (cfg.log_ranks << 16), (const u32 *)cfg.omega);
    PANDA_TRY(tw.settle(stream));
    int dev = -1;
    PANDA_TRY(hipGetDevice(&dev));
    const bool hit = tw.valid && tw.device == dev && memcmp(key, tw.key, sizeof(key)) == 0;
    if (!hit)
        PANDA_TRY(tw.ensure(SZ_PQ + 4096));
    else
        tw.used = 0;
    tw.valid = false;
    u32 *d_pq = (u32 *)tw.take(SZ_PQ);
    if (!d_pq) return hipErrorOutOfMemory;
    if (!hit) {
        Fe<Fr> omega, base;
        fe_from_wire(omega, (const u32 *)cfg.omega);
        if (inverse) {
            Fe<Fr> oi;
            fe_inv(oi, omega);
            omega = oi;
        }
        fe_pow_u64(base, omega, m); // w^m (w^-m for the inverse) has order G
        if (cfg.log_ranks <= 3)
            build_table2<Fr>(stream, base, nullptr, std::max(1u, (1u << cfg.log_ranks) >> 1), d_pq);
        else
            build_table<Fr>(stream, base, nullptr, std::max(1u, (1u << cfg.log_ranks) >> 1), d_pq);
    }
    if (cfg.log_ranks <= 3) { // 2, 4 or 8 ranks: the size-G transforms run in registers, one thread per column of the G x m/G slab
        launch_small<Fr>(cfg.log_ranks, (const u32 *)cfg.d_slab, (u32 *)cfg.d_scratch, d_pq, log_m - cfg.log_ranks, false, stream);
        PANDA_TRY(hipGetLastError());
        if (cfg.flag) *(unsigned *)cfg.flag = 1;
        return slab_finish(tw, key, stream, wait);
    }
    PassArgs a{};
    a.x = (const u32 *)cfg.d_slab;
    a.y = (u32 *)cfg.d_scratch;
    a.pq = d_pq;
    a.ta = a.tb = d_pq;
    a.log_n = log_m;
    a.lgp = 0;
    a.split = 0;
    a.force_tw = 0;
    a.tile_elems = (unsigned)std::min<u64>(TILE, m);
    a.strided_out = 1;
    a.canonical = 1;
    launch_pass<Fr>(cfg.log_ranks, a, (unsigned)(m / a.tile_elems), stream);
    PANDA_TRY(hipGetLastError());
    if (cfg.flag) *(unsigned *)cfg.flag = 1;
    return slab_finish(tw, key, stream, wait);
}

// Inverse of the sharded transform, second half (after the exchange): the slab holds B_r[k2]; multiply by w^(-r k2) / n, then the
// local size-m inverse transform (root w^-G).  The mirror image of slab_step1: twiddle first, passes second.
template <class Fr>
hipError_t slab_inverse_local(const panda_ntt_slab_configuration &cfg, bool wait)
{
    if (cfg.log_ranks > 8 || cfg.log_n > 28 || cfg.log_n < cfg.log_ranks || !cfg.d_slab || !cfg.d_scratch || !cfg.omega) return hipErrorInvalidValue;
    if (cfg.rank >= (1u << cfg.log_ranks)) return hipErrorInvalidValue;
    if (slab_too_short(cfg)) return hipErrorInvalidValue;
    hipStream_t stream = static_cast<hipStream_t>(cfg.stream.handle);
    PANDA_TRY(order_after_null_stream(stream));
    const unsigned log_m = cfg.log_n - cfg.log_ranks;
    const u64 m = (u64)1 << log_m;
    TwiddleCache &tw = g_twiddles[TW_SLAB1];
    u32 key[12];
    twiddle_key<Fr>(key, cfg.log_n, 0x300u | (cfg.log_ranks << 16) | (cfg.rank << 20), (const u32 *)cfg.omega);
    PANDA_TRY(tw.settle(stream));
    int dev = -1;
    PANDA_TRY(hipGetDevice(&dev));
    const bool hit = tw.valid && tw.device == dev && memcmp(key, tw.key, sizeof(key)) == 0;
    Fe<Fr> omega_inv, omega_m, scale;
    fe_zero(omega_inv);
    fe_zero(omega_m);
    fe_zero(scale);
    if (!hit) {
        fe_from_wire(omega_inv, (const u32 *)cfg.omega);
        inverse_parameters<Fr>(omega_inv, scale, (u64)1 << cfg.log_n); // w^-1 and n^-1
        fe_pow_u64(omega_m, omega_inv, (u64)1 << cfg.log_ranks);
        PANDA_TRY(tw.ensure(passes_table_bytes(log_m) + SZ_TA + SZ_TB + 4096));
    } else
        tw.used = 0;
    tw.valid = false;
    u32 *d_ta = (u32 *)tw.take(SZ_TA), *d_tb = (u32 *)tw.take(SZ_TB);
    if (!d_ta || !d_tb) return hipErrorOutOfMemory;
    if (!hit) {
        Fe<Fr> base, base_b;
        fe_pow_u64(base, omega_inv, cfg.rank);
        build_table<Fr>(stream, base, &scale, (unsigned)std::min<u64>(m, 1u << 16), d_ta); // n^-1 folded into the low table
        if (log_m > 16) {
            fe_pow_u64(base_b, base, (u64)1 << 16);
            build_table<Fr>(stream, base_b, nullptr, 1u << (log_m - 16), d_tb);
        }
    }
    hipLaunchKernelGGL(k_slab_twiddle<Fr>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, (u32 *)cfg.d_slab, d_ta, d_tb, (unsigned)m);
    PANDA_TRY(hipGetLastError());
    unsigned passes = 0;
    PANDA_TRY(ntt_passes<Fr>(stream, tw, (const u32 *)cfg.d_slab, (u32 *)cfg.d_scratch, omega_m, log_m, nullptr, &passes, !hit));
    if (cfg.flag) *(unsigned *)cfg.flag = passes & 1u;
    return slab_finish(tw, key, stream, wait);
}

} // namespace

extern "C" {

panda_error panda_ntt_setup_bn254(void *input_omega)
{
    if (!input_omega) return panda_error_invalid_value;
    std::lock_guard<std::mutex> lock(g_omega_mutex);
    memcpy(g_omega_wire, input_omega, sizeof(g_omega_wire));
    g_omega_set = true;
    return panda_success;
}

panda_error panda_ntt_execute_bn254(panda_ntt_configuration cfg)
{
    u32 omega[8];
    {
        std::lock_guard<std::mutex> lock(g_omega_mutex);
        if (!g_omega_set) return panda_error_invalid_value;
        memcpy(omega, g_omega_wire, sizeof(omega));
    }
    return static_cast<panda_error>(ntt_run<Bn254Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, omega, cfg.log_n, (unsigned *)cfg.flag, false));
}

panda_error panda_ntt_execute_bn254_v1(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bn254Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, false));
}

panda_error panda_ntt_execute_bn254_inverse(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bn254Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, true));
}

// Orderings a prover's polynomial pipeline chains without ever materialising the natural order in between: the forward
// transform leaves y[k] at bitrev(k) (the last pass copies its LDS tile straight out), the inverse takes such a buffer
// (the first pass reads contiguous runs) and returns natural-order coefficients, n^-1 fused.  Same passes, same flag protocol.
panda_error panda_ntt_execute_bn254_bitrev_out(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bn254Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, false, false, true));
}

panda_error panda_ntt_execute_bn254_inverse_bitrev_in(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bn254Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, true, true, false));
}

panda_error panda_ntt_slab_step1_bn254(const panda_ntt_slab_configuration cfg) { return static_cast<panda_error>(slab_step1<Bn254Fr>(cfg, true)); }

panda_error panda_ntt_slab_step2_bn254(const panda_ntt_slab_configuration cfg) { return static_cast<panda_error>(slab_step2<Bn254Fr>(cfg, true)); }

// The same two steps without the wait at the end: everything is enqueued on cfg.stream and *flag is written before the call returns
// (it is the parity of the pass count), so a caller that issues the all-to-all on the same stream composes
// step 1 -> exchange -> step 2 with a single synchronisation at the end instead of three.
panda_error panda_ntt_slab_step1_bn254_enqueue(const panda_ntt_slab_configuration cfg) { return static_cast<panda_error>(slab_step1<Bn254Fr>(cfg, false)); }

panda_error panda_ntt_slab_step2_bn254_enqueue(const panda_ntt_slab_configuration cfg) { return static_cast<panda_error>(slab_step2<Bn254Fr>(cfg, false)); }

// Inverse of the sharded transform: it takes the forward transform's OUTPUT layout (rank q: y[k1 m + q m/G + k2'] at [k1][k2']) back to
// its INPUT layout (rank r: x[r + G j2]), n^-1 included, by running the steps backwards: size-G inverse transforms down k1
// (inverse_step1), the same all-to-all, then the twiddle w^(-r k2) / n and the local size-m inverse transform (inverse_step2).
// cfg.omega is the FORWARD root.  Enqueued on cfg.stream without waiting; flag as for the forward steps.
panda_error panda_ntt_slab_inverse_step1_bn254_enqueue(const panda_ntt_slab_configuration cfg) { return static_cast<panda_error>(slab_step2<Bn254Fr>(cfg, false, true)); }

panda_error panda_ntt_slab_inverse_step2_bn254_enqueue(const panda_ntt_slab_configuration cfg) { return static_cast<panda_error>(slab_inverse_local<Bn254Fr>(cfg, false)); }

panda_error panda_ntt_execute_bn254_coset(const panda_ntt_configuration_v1 cfg, const void *shift)
{
    return static_cast<panda_error>(ntt_coset_run<Bn254Fr>(cfg, shift, false));
}

panda_error panda_ntt_execute_bn254_coset_inverse(const panda_ntt_configuration_v1 cfg, const void *shift)
{
    return static_cast<panda_error>(ntt_coset_run<Bn254Fr>(cfg, shift, true));
}

// BLS12-377 Fr (two-adicity 47): same kernels, other field parameters
panda_error panda_ntt_execute_bls12_377_v1(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bls377Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, false));
}

panda_error panda_ntt_execute_bls12_377_inverse(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bls377Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, true));
}

// ... in every variant the BN254 field has (north_star: "NTT butterfly over BN254 / BLS12-377"): bit-reversed orderings, coset
// transforms and the two halves of the slab-sharded transform (panda_ntt_execute_bls12_377[_inverse]_multi composes them, multi_gpu.hip)
panda_error panda_ntt_execute_bls12_377_bitrev_out(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bls377Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, false, false, true));
}

panda_error panda_ntt_execute_bls12_377_inverse_bitrev_in(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bls377Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, true, true, false));
}

panda_error panda_ntt_execute_bls12_377_coset(const panda_ntt_configuration_v1 cfg, const void *shift)
{
    return static_cast<panda_error>(ntt_coset_run<Bls377Fr>(cfg, shift, false));
}

panda_error panda_ntt_execute_bls12_377_coset_inverse(const panda_ntt_configuration_v1 cfg, const void *shift)
{
    return static_cast<panda_error>(ntt_coset_run<Bls377Fr>(cfg, shift, true));
}

panda_error panda_ntt_slab_step1_bls12_377_enqueue(const panda_ntt_slab_configuration cfg) { return static_cast<panda_error>(slab_step1<Bls377Fr>(cfg, false)); }
panda_error panda_ntt_slab_step2_bls12_377_enqueue(const panda_ntt_slab_configuration cfg) { return static_cast<panda_error>(slab_step2<Bls377Fr>(cfg, false)); }
panda_error panda_ntt_slab_inverse_step1_bls12_377_enqueue(const panda_ntt_slab_configuration cfg)
{
    return static_cast<panda_error>(slab_step2<Bls377Fr>(cfg, false, true));
}
panda_error panda_ntt_slab_inverse_step2_bls12_377_enqueue(const panda_ntt_slab_configuration cfg)
{
    return static_cast<panda_error>(slab_inverse_local<Bls377Fr>(cfg, false));
}

// BLS12-381 Fr (two-adicity 32)
panda_error panda_ntt_execute_bls12_381_v1(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bls381Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, false));
}

panda_error panda_ntt_execute_bls12_381_inverse(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bls381Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, true));
}

// ... and the same variants over BLS12-381 Fr ("more curves / NTT variants")
panda_error panda_ntt_execute_bls12_381_bitrev_out(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bls381Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, false, false, true));
}

panda_error panda_ntt_execute_bls12_381_inverse_bitrev_in(const panda_ntt_configuration_v1 cfg)
{
    return static_cast<panda_error>(
        ntt_run<Bls381Fr>(static_cast<hipStream_t>(cfg.stream.handle), cfg.d_src, cfg.d_dst, (const u32 *)cfg.d_omega, cfg.log_n, (unsigned *)cfg.flag, true, true, false));
}

panda_error panda_ntt_execute_bls12_381_coset(const panda_ntt_configuration_v1 cfg, const void *shift)
{
    return static_cast<panda_error>(ntt_coset_run<Bls381Fr>(cfg, shift, false));
}

panda_error panda_ntt_execute_bls12_381_coset_inverse(const panda_ntt_configuration_v1 cfg, const void *shift)
{
    return static_cast<panda_error>(ntt_coset_run<Bls381Fr>(cfg, shift, true));
}

panda_error panda_ntt_slab_step1_bls12_381_enqueue(const panda_ntt_slab_configuration cfg) { return static_cast<panda_error>(slab_step1<Bls381Fr>(cfg, false)); }
panda_error panda_ntt_slab_step2_bls12_381_enqueue(const panda_ntt_slab_configuration cfg) { return static_cast<panda_error>(slab_step2<Bls381Fr>(cfg, false)); }
panda_error panda_ntt_slab_inverse_step1_bls12_381_enqueue(const panda_ntt_slab_configuration cfg)
{
    return static_cast<panda_error>(slab_step2<Bls381Fr>(cfg, false, true));
}
panda_error panda_ntt_slab_inverse_step2_bls12_381_enqueue(const panda_ntt_slab_configuration cfg)
{
    return static_cast<panda_error>(slab_inverse_local<Bls381Fr>(cfg, false));
}

panda_error panda_ntt_last_clock(uint64_t *out)
{
    if (!out) return panda_error_invalid_value;
    panda::clock_delta_out(panda::thread_ntt_clock(), out);
    return panda_success;
}

panda_error panda_ntt_last_device_ms(float *ms)
{
    if (!ms) return panda_error_invalid_value;
    *ms = g_pass_timer.ms;
    return panda_success;
}

panda_error panda_ntt_set_streamed_tables(unsigned mode)
{
    if (mode == 0xffffffffu) mode = STREAMED_POLICY; // "restore the built-in policy"
    if (mode > STREAMED_FAIL_ALLOC) return panda_error_invalid_value;
    g_streamed_tables.store(mode, std::memory_order_relaxed);
    g_streamed_unavailable = StreamedUnavailable{}; // of the calling thread: a new setting gets a new try
    return panda_success;
}

panda_error panda_ntt_table_builds(uint64_t *count)
{
    if (!count) return panda_error_invalid_value;
    *count = g_table_builds;
    return panda_success;
}

panda_error panda_ntt_pass_plan(unsigned log_n, unsigned *passes, unsigned *radix_bits)
{
    if (log_n > 28 || !passes) return panda_error_invalid_value;
    const PassPlan pl = plan_passes(log_n);
    *passes = pl.count;
    if (radix_bits)
        for (unsigned j = 0; j < 4; j++) radix_bits[j] = j < pl.count ? pl.d[j] : 0;
    return panda_success;
}

panda_error panda_ntt_tear_down(void)
{
    std::lock_guard<std::mutex> lock(g_omega_mutex);
    g_omega_set = false;
    g_streamed_unavailable = StreamedUnavailable{};
    for (auto &t : g_twiddles) (void)t.release();
    return static_cast<panda_error>(panda::release_thread_arena());
}

} // extern "C"
