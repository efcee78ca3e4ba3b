// ntt_radix8.h -- one radix-256 pass of the transform with the data held in registers (included by ntt.hip).
//
// The pass keeps the reference protocol's shape (src/cuda/core/unit/ntt/fft.cu:171-216: a radix-2^8 pass from one buffer to the
// other) but not the one-butterfly-per-thread-per-round layout of k_ntt_pass:
//
//   * a workgroup of 256 threads owns a tile of 8 sub-transforms (2048 elements); every thread owns EIGHT elements and runs
//     three butterfly rounds on them in registers, so the 8 rounds are three register blocks (3 + 3 + 2 rounds) with two
//     exchanges through LDS in between instead of eight LDS round trips with a barrier each.  The elements come from HBM
//     straight into registers and leave from registers (the thread-to-element maps of the first and last block are chosen so
//     that a wave touches runs of 256 B or more);
//   * every product has a constant factor (a twiddle), so it is a precomputed-quotient product (fe_mul_shoup, fe29.h):
//     143 multiply-adds instead of 171, no radix factor.  Tables hold (w, floor(w R / p)) pairs;
//   * in the middle block the twiddle of a butterfly depends only on the wave and the register index, so it lives in scalar
//     registers, and the waves whose twiddle is 1 skip the product; the last block's only twiddle is the 4th root of unity;
//   * the twiddle between two passes moved from the input of the later pass to the OUTPUT of the earlier one: the product
//     takes the un-reduced output of the last round (anything below R) and returns a value below 3p, which fits the 32-byte
//     wire element -- it doubles as the reduction the old kernel ran separately.  Only the last pass of a transform, which
//     has no twiddle, reduces (to the canonical range);
//   * LDS is an exchange medium only.  The limbs of a tile travel in batches of PB limb planes (PB * 8 KB per workgroup) so that
//     three or four workgroups fit a CU next to the registers they need; XOR-swizzled addresses keep both sides of an
//     exchange free of bank conflicts.
#pragma once
#include <hip/hip_runtime.h>

#include "fe29.h"

namespace panda_ntt8 {
using namespace panda29;

constexpr int NL = 9;
constexpr int TW2_STRIDE = 20; // words per table entry: w[9], floor(w R / p)[9], two words of padding (80 B, five 16-byte loads)
constexpr int SUBS = 8;        // sub-transforms per tile
constexpr int ELEMS = 2048;    // elements per tile
constexpr int THREADS = 256;

struct Pass8Args {
    const u32 *x;
    u32 *y;
    const u32 *pq; // 128 entries: w_256^t, the butterfly twiddles of a 256-point sub-transform
    const u32 *ta; // output twiddles, see below
    const u32 *tb;
    unsigned log_n;
    unsigned lgp;      // log2 of the product of the radices of the passes before this one
    unsigned ca;       // the twiddle of output element (blk, i_out) is W^(i2 k2), k2 = i_out 2^lgp + (blk mod 2^lgp) its index within
    unsigned cb;       // the next pass's sub-transform group and i2 = (blk >> lgp) >> i2_shift its input index there.  cb == 0: one
    unsigned i2_shift; // lookup ta[(i2 << ca) | k2].  Else two: ta[(i2 << ca) | (k2 mod 2^ca)] (ca <= lgp: one per thread) and
                       // tb[(i2 << cb) | (k2 >> ca)]
    unsigned br_in;    // first pass: the caller's input is in bit-reversed order
    unsigned br_out;   // last pass: leave the output in bit-reversed order
    unsigned tiles;    // tiles of the pass (a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...)
    unsigned wide;     // not the last pass, cb != 0 geometry: the output twiddle W^(i2 k2) comes from ONE table of 2^(deg2 + lgp + 8) entries -- ta, 32 bytes
                       // each: the twiddle times the kernels' Montgomery radix, in wire limbs -- streamed beside the data ([i2][k2]: a thread's entries
                       // sit where its outputs sit), and the product is one ordinary Montgomery product instead of two precomputed-quotient ones
    unsigned skip;     // LAST pass only: a pass of radix 2^(8 - skip), skip = 1 .. 4.  The top `skip` bits of the 8-bit local index then
                       // select the sub-transform instead of a position in it (a tile holds 8 << skip sub-transforms) and rounds
                       // 0 .. skip - 1 do not run; the twiddle indices of the remaining rounds are what they were (powers of the
                       // 256-th root w^(n/256) that are multiples of 2^skip are the powers of the 2^(8-skip)-th root).
};

template <class Fr>
struct TwV {
    u32 w[NL];
    u32 q[NL];
};

template <class Fr>
__device__ __forceinline__ void tw_from_quads(TwV<Fr> &t, const uint4 &a, const uint4 &b, const uint4 &c, const uint4 &d, const uint4 &e)
{
    t.w[0] = a.x; t.w[1] = a.y; t.w[2] = a.z; t.w[3] = a.w;
    t.w[4] = b.x; t.w[5] = b.y; t.w[6] = b.z; t.w[7] = b.w;
    t.w[8] = c.x;
    t.q[0] = c.y; t.q[1] = c.z; t.q[2] = c.w;
    t.q[3] = d.x; t.q[4] = d.y; t.q[5] = d.z; t.q[6] = d.w;
    t.q[7] = e.x; t.q[8] = e.y;
}

template <class Fr>
__device__ __forceinline__ void load_tw2(TwV<Fr> &t, const u32 *tab, unsigned idx)
{
    const uint4 *p = reinterpret_cast<const uint4 *>(tab + (size_t)idx * TW2_STRIDE);
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    tw_from_quads(t, a, b, c, d, e);
}

// wave-uniform entry: word loads from a uniform address, which the compiler turns into scalar loads
template <class Fr>
__device__ __forceinline__ void load_tw2_uniform(TwV<Fr> &t, const u32 *__restrict__ tab, unsigned idx)
{
    const u32 *p = tab + idx * TW2_STRIDE;
#pragma unroll
    for (int j = 0; j < NL; j++) {
        t.w[j] = __builtin_amdgcn_readfirstlane(p[j]);
        t.q[j] = __builtin_amdgcn_readfirstlane(p[NL + j]);
    }
}

// ---- compile-time bounds ------------------------------------------------------------------------------------------------------
// All elements of a pass carry one bound (in units of p) per round.  A butterfly on inputs below B p stores a + b (< 2B p) and
// either (a - b + K p) w (< 3p out of the product, which needs a - b + K p < R) or, where the twiddle is 1, a - b + K p itself.
// When the next round could not take the result, this round reduces its un-multiplied outputs below 2p (fe_reduce_mad_2p).
template <class Fr>
constexpr int keff_of(int B)
{
    return Fr::KEFF[B + SubMargin<Fr>::value];
}
template <class Fr>
constexpr bool round_ok(int B)
{
    return B + SubMargin<Fr>::value <= 200 && B + keff_of<Fr>(B) < (int)Fr::HEADROOM;
}
template <class Fr, int B0>
struct Plan8 {
    int b[9];
    bool red[8];
    constexpr Plan8() : b{}, red{}
    {
        int B = B0;
        for (int r = 0; r < 8; r++) {
            b[r] = B;
            const int dplain = B + keff_of<Fr>(B);
            const int dmax = (r >= 3 && dplain > 3) ? dplain : 3; // rounds 0..2 multiply every difference
            int next = 2 * B > dmax ? 2 * B : dmax;
            red[r] = r < 7 ? !round_ok<Fr>(next) : next >= (int)Fr::HEADROOM; // the output product wants its operand below R
            if (red[r]) next = 3;
            B = next;
        }
        b[8] = B;
    }
};

template <class Fr, int B0>
inline constexpr Plan8<Fr, B0> plan8_v{};

// One radix-2 butterfly on registers: a <- a + b, b <- (a - b + K p) [* w].  Carry passes are spent where the limbs need them, not
// after every addition (MODE by round parity):
//   MODE 0 (rounds 0, 2, 4, 6)  inputs tight or loose; the sum is left un-normalised (limbs <= 2^30 + 16)
//   MODE 1 (rounds 1, 3, 5)     inputs may be such sums; the difference takes the 3 * 2^29 bias that covers them and goes into the
//                               product as it is (limbs < 3 * 2^30 + 32, what fe_mul_shoup's columns hold); the sum is normalised
//   MODE 2 (round 7)            inputs as MODE 1; both outputs stay un-normalised: they go straight into the output product or reduction
template <class Fr, int B, bool MULT, bool RED, bool UNIFORM, int MODE>
__device__ __forceinline__ void bfly(Fe<Fr> &a, Fe<Fr> &b, const u32 *w, const u32 *wq, bool unit = false)
{
    static_assert(B + SubMargin<Fr>::value <= 200, "subtraction constant table too small");
    static_assert(!(MODE == 2 && (MULT || RED)), "the last round only adds and subtracts");
    Fe<Fr> s, d;
    fe_add_nr(s, a, b);
    if constexpr (MULT) {
        static_assert(round_ok<Fr>(B), "operand of the twiddle product must stay below R");
        Fe<Fr> x;
        fe_sub_raw_bias<Fr, B, MODE == 0 ? 2 : 3>(x, a, b);
        if (unit) { // wave-uniform: this wave's twiddle is 1 -- the difference only needs its carries
            fe_norm(d, x);
            if constexpr (RED) fe_reduce_mad_2p(d);
        } else
            fe_mul_shoup<Fr, UNIFORM>(d, x, w, wq);
    } else if constexpr (MODE == 2)
        fe_sub_raw_bias<Fr, B, 3>(d, a, b);
    else {
        fe_sub<Fr, B>(d, a, b); // limbs of b < 2^31 - 4: fine for either input class
        if constexpr (RED) fe_reduce_mad_2p(d);
    }
    if constexpr (RED)
        fe_reduce_mad_2p(s);
    else if constexpr (MODE == 1)
        fe_norm(s, s);
    a = s;
    b = d;
}

__device__ __forceinline__ unsigned brev(unsigned v, unsigned bits) { return __brev(v) >> (32 - bits); } // bits >= 1
__device__ __forceinline__ unsigned brev0(unsigned v, unsigned bits) { return bits ? __brev(v) >> (32 - bits) : 0u; }
constexpr unsigned br3(unsigned m) { return ((m & 1) << 2) | (m & 2) | ((m >> 2) & 1); }

template <class Fr>
__device__ __forceinline__ void load_elem32(Fe<Fr> &v, const u32 *__restrict__ src)
{
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    const uint4 lo = s4[0], hi = s4[1];
    const u32 w8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    fe_unpack(v, w8);
}
template <class Fr>
__device__ __forceinline__ void store_elem32(u32 *__restrict__ dst, const Fe<Fr> &v) // tight limbs, value < 2^256
{
    u32 w8[8];
    fe_pack(w8, v);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    d4[0] = make_uint4(w8[0], w8[1], w8[2], w8[3]);
    d4[1] = make_uint4(w8[4], w8[5], w8[6], w8[7]);
}

// Exchange through LDS, PB limb planes at a time: every thread stores limb planes [p0, p0 + PB) of its eight elements at wa[m],
// the workgroup meets, every thread loads the same planes of its eight NEW elements from ra + m * RSTEP.  A plane is ELEMS words.
template <class Fr, int PB, int RSTEP>
__device__ __forceinline__ void exchange(Fe<Fr> (&e)[8], u32 *s_x, const unsigned (&wa)[8], unsigned ra)
{
    Fe<Fr> n[8];
#ifdef P8_PRIO // experiment switch: exchange phases at raised wave priority (a workgroup parked at a barrier holds four waves' registers)
    __builtin_amdgcn_s_setprio(2);
#endif
#pragma unroll
    for (int p0 = 0; p0 < NL; p0 += PB) {
        if (p0 != 0) __syncthreads(); // the previous batch has been read by everyone
#pragma unroll
        for (int m = 0; m < 8; m++)
#pragma unroll
            for (int pl = p0; pl < p0 + PB && pl < NL; pl++) s_x[(pl - p0) * ELEMS + wa[m]] = e[m].l[pl];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; m++)
#pragma unroll
            for (int pl = p0; pl < p0 + PB && pl < NL; pl++) n[m].l[pl] = s_x[(pl - p0) * ELEMS + ra + m * RSTEP];
    }
#pragma unroll
    for (int m = 0; m < 8; m++) e[m] = n[m];
#ifdef P8_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// FIRST: input bound 2 (the caller's elements; nothing was multiplied onto them), else 3 (outputs of the previous pass's twiddle
// product).  LAST: no output twiddle, canonical output.
// experiment switch (tools/ntt8_variants.hip): the middle block's twiddles from the LDS copy of the table instead of scalar loads
#ifdef P8_B_LDS
#define P8_LOAD_B(t, idx) load_tw2(t, s_tw, idx)
#define P8_B_UNIFORM false
#else
#define P8_LOAD_B(t, idx) load_tw2_uniform(t, A.pq, idx)
#define P8_B_UNIFORM true
#endif

template <class Fr, bool FIRST, bool LAST, int PB, int MINW>
__global__ void __launch_bounds__(THREADS, MINW) k_ntt_pass8(Pass8Args A)
{
    constexpr const Plan8<Fr, FIRST ? 2 : 3> &PL = plan8_v<Fr, FIRST ? 2 : 3>;
    __shared__ u32 s_x[PB * ELEMS];
    __shared__ __attribute__((aligned(16))) u32 s_tw[128 * TW2_STRIDE];

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned skip = LAST ? A.skip : 0u, deg = 8 - skip;
    const unsigned S = (1u << A.log_n) >> deg; // stride between the inputs of one sub-transform
    const unsigned lgp = A.lgp;

    // butterfly twiddles of block A -> LDS (2560 words, ten per thread)
    {
        const uint2 *g = reinterpret_cast<const uint2 *>(A.pq) + tid * 5;
        uint2 *l = reinterpret_cast<uint2 *>(s_tw) + tid * 5;
#pragma unroll
        for (int j = 0; j < 5; j++) l[j] = g[j];
    }
#ifdef P8_PERSIST // experiment switch: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (costs registers: measured slower)
#pragma unroll 1
  for (unsigned tile = blockIdx.x; tile < A.tiles; tile += gridDim.x) {
#else
  {
    const unsigned tile = blockIdx.x;
#endif
    const unsigned blk0 = tile * (SUBS << skip);

    // ---- block A: thread (s, i0) holds i = i0 + 32 m of sub-transform s; rounds 0..2 (distances 128, 64, 32)
    Fe<Fr> e[8];
    unsigned s, i0;
    {
        size_t base, step;
        if (A.br_in) {
            // element j = blk + i S of the natural order sits at bitrev(j) = (bitrev(blk) << 8) + bitrev8(i): a sub-transform is one
            // contiguous run, a thread's eight elements (i = i0 + 32 m -> bitrev5(i0) * 8 + bitrev3(m)) are 256 contiguous bytes
            const unsigned bi = lane & 31;
            i0 = brev(bi, 5);
            s = (lane >> 5) | (wave << 1);
            base = ((size_t)brev0(blk0 + s, A.log_n - 8) << 8) + (bi << 3);
#pragma unroll
            for (int m = 0; m < 8; m++) load_elem32(e[m], A.x + (base + br3(m)) * 8);
        } else if (skip == 0) {
            s = lane & 7;
            i0 = (lane >> 3) | (wave << 3);
            base = (size_t)(blk0 + s) + (size_t)i0 * S;
            step = (size_t)32 * S;
#pragma unroll
            for (int m = 0; m < 8; m++) load_elem32(e[m], A.x + (base + m * step) * 8);
        } else { // local index i8 = i0 + 32 m: its top `skip` bits pick the sub-transform (8 apart), the rest the position
            s = lane & 7;
            i0 = (lane >> 3) | (wave << 3);
            const unsigned mask = (1u << deg) - 1;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const unsigned i8 = i0 + 32 * m;
                load_elem32(e[m], A.x + ((size_t)(blk0 + ((i8 >> deg) << 3) + s) + (size_t)(i8 & mask) * S) * 8);
            }
        }
    }
    __syncthreads(); // s_tw complete (first tile); the previous tile's last exchange batch has been read by everyone (later ones)

    {
        TwV<Fr> t;
        // round 0: pairs (m, m + 4), twiddle index i0 + 32 m
        if (skip < 1) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                load_tw2(t, s_tw, i0 + 32 * m);
                bfly<Fr, PL.b[0], true, PL.red[0], false, 0>(e[m], e[m + 4], t.w, t.q);
            }
        }
        // round 1: pairs (m, m + 2), twiddle index 2 (i0 + 32 (m & 1))
        if (skip < 2) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                load_tw2(t, s_tw, 2 * (i0 + 32 * h));
                bfly<Fr, PL.b[1], true, PL.red[1], false, 1>(e[h], e[h + 2], t.w, t.q);
                bfly<Fr, PL.b[1], true, PL.red[1], false, 1>(e[h + 4], e[h + 6], t.w, t.q);
            }
        }
        // round 2: pairs (m, m + 1), twiddle index 4 i0
        if (skip < 3) {
            load_tw2(t, s_tw, 4 * i0);
#pragma unroll
            for (int m = 0; m < 8; m += 2) bfly<Fr, PL.b[2], true, PL.red[2], false, 0>(e[m], e[m + 1], t.w, t.q);
        }
    }

    // ---- exchange 1: element (s, i) lives at word s | ((i[7:5] ^ i[1:0]) << 3) | (i[4:0] << 6) of each plane
    {
        unsigned wa[8];
        const unsigned wbase = s | ((i0 & 3) << 3) | (i0 << 6);
#pragma unroll
        for (int m = 0; m < 8; m++) wa[m] = wbase ^ (m << 3);
        // block B: thread (s, g, j0) holds i = 32 g + 4 m + j0; j0 is the wave
        s = lane & 7;
        const unsigned g = lane >> 3, j0 = wave;
        const unsigned ra = s | ((g ^ j0) << 3) | (j0 << 6);
        exchange<Fr, PB, 256>(e, s_x, wa, ra);
    }

    // ---- block B: rounds 3..5 (distances 16, 8, 4).  Twiddle indices 8 (4 (m & 3) + j0), 16 (4 (m & 1) + j0), 32 j0: wave-uniform
    // In wave 0 the twiddle is 1 wherever the register bits below the round's own are clear (7 of its 12 butterflies): those skip the
    // product.  One code path with a wave-uniform test per butterfly -- as two separate paths the compiler hoisted their common sums
    // and differences above the branch and spilled 68 registers there.
    {
        const unsigned j0 = wave;
        const bool w0 = j0 == 0;
        TwV<Fr> t;
        if (skip < 4) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                P8_LOAD_B(t, 8 * (4 * m + j0));
                bfly<Fr, PL.b[3], true, PL.red[3], P8_B_UNIFORM, 1>(e[m], e[m + 4], t.w, t.q, m == 0 && w0);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            P8_LOAD_B(t, 16 * (4 * h + j0));
            bfly<Fr, PL.b[4], true, PL.red[4], P8_B_UNIFORM, 0>(e[h], e[h + 2], t.w, t.q, h == 0 && w0);
            bfly<Fr, PL.b[4], true, PL.red[4], P8_B_UNIFORM, 0>(e[h + 4], e[h + 6], t.w, t.q, h == 0 && w0);
        }
        P8_LOAD_B(t, 32 * j0);
#pragma unroll
        for (int m = 0; m < 8; m += 2) bfly<Fr, PL.b[5], true, PL.red[5], P8_B_UNIFORM, 1>(e[m], e[m + 1], t.w, t.q, w0);
    }

    // ---- exchange 2: element (s, i) lives at word b | (s << 5) | (i[2:0] << 8), b[1:0] = i[6:5] ^ i[4:3], b[4:2] = s ^ (i[7], i[4:3])
    unsigned q;
    {
        __syncthreads(); // everyone has read exchange 1's last batch
        unsigned wa[8];
        const unsigned g = lane >> 3, j0 = wave; // s is still lane & 7
        const unsigned wbase = (g & 3) | ((s ^ (g & 4)) << 2) | (s << 5) | (j0 << 8);
#pragma unroll
        for (int m = 0; m < 8; m++) wa[m] = (wbase ^ ((m >> 1) * 5)) + ((m & 1) << 10); // i[4:3] = m >> 1 enters both fields, i[2] = m & 1
        // block C: thread (s, q) holds i = 8 q + m
        if (A.br_out) {
            q = lane & 31;
            s = (lane >> 5) | (wave << 1);
        } else if (lgp == 0) { // outputs i_out = bitrev8(8 q + m) = 32 bitrev3(m) + bitrev5(q) of one sub-transform are contiguous in i_out
            q = brev(lane & 31, 5);
            s = (lane >> 5) | (wave << 1);
        } else { // consecutive sub-transforms are contiguous
            s = lane & 7;
            q = (lane >> 3) | (wave << 3);
        }
        const unsigned ra = (((q >> 2) ^ q) & 3) | ((s ^ (((q >> 4) << 2) | (q & 3))) << 2) | (s << 5);
        exchange<Fr, PB, 256>(e, s_x, wa, ra);
    }

    // ---- block C: rounds 6, 7 (distances 2, 1).  Round 6: twiddle 1 for even i, the 4th root of unity w^64 for odd i; round 7: 1
    {
        TwV<Fr> t;
        load_tw2_uniform(t, A.pq, 64);
        bfly<Fr, PL.b[6], false, PL.red[6], true, 0>(e[0], e[2], nullptr, nullptr);
        bfly<Fr, PL.b[6], false, PL.red[6], true, 0>(e[4], e[6], nullptr, nullptr);
        bfly<Fr, PL.b[6], true, PL.red[6], true, 0>(e[1], e[3], t.w, t.q);
        bfly<Fr, PL.b[6], true, PL.red[6], true, 0>(e[5], e[7], t.w, t.q);
#pragma unroll
        for (int m = 0; m < 8; m += 2) bfly<Fr, PL.b[7], false, PL.red[7], true, PL.red[7] ? 1 : 2>(e[m], e[m + 1], nullptr, nullptr);
    }
    constexpr int FB = PL.b[8];
    static_assert(FB < (int)Fr::HEADROOM && FB < 512, "final bound");
    // the output product of a pass that is not the last is stored as a 32-byte element: its value, below (1 + x / R + 2^-23) p with
    // x < FB p (fe29.h, fe_mul_shoup), must stay below 2^(32 L).  p < (PW[L-1] + 1) 2^(32 (L-1)) and x / R < FB / HEADROOM:
    static_assert(LAST || ((unsigned long long)(Fr::HEADROOM + FB + 1) * ((unsigned long long)Fr::PW[Fr::L - 1] + 1) < (unsigned long long)Fr::HEADROOM << 32),
                  "an inter-pass output product would not fit the 32-byte element");

    // ---- output: register m holds output i_out = 32 bitrev3(m) + bitrev5(q) of sub-transform blk = blk0 + s
    const unsigned blk = blk0 + s;
    const unsigned p = 1u << lgp, k = blk & (p - 1);
    const unsigned iq = brev(q, 5);
    if constexpr (LAST) {
        // the last pass has blk < 2^lgp: y[blk + i_out 2^lgp], or bit-reversed the run of sub-transform blk at bitrev(blk) 2^deg
        const unsigned smask = (1u << skip) - 1, pmask = (1u << deg) - 1;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            fe_reduce_mad_2p(e[m]);
            fe_reduce_once(e[m]);
            const unsigned v = (br3(m) << 5) | iq;                        // bitrev8 of the local index 8 q + m
            const unsigned sub = blk0 + (brev0(v & smask, skip) << 3) + s; // its top `skip` bits, un-reversed, pick the sub-transform
            const size_t dst = A.br_out ? ((size_t)brev0(sub, lgp) << deg) + ((8 * q + m) & pmask) : (size_t)sub + ((size_t)(v >> skip) << lgp);
            store_elem32(A.y + dst * 8, e[m]);
        }
    } else {
        const size_t base = ((size_t)(blk - k) << 8) + k + ((size_t)iq << lgp);
        const unsigned i2 = (blk >> lgp) >> A.i2_shift;
        // The table entries of the eight products are fetched one product ahead and no further (a scheduling barrier per element):
        // hoisted all at once they are 160 registers, and the spills that buys showed up as 0.8 GB of scratch traffic per launch
        if (A.wide) { // one streamed table over the whole index range: entry (i2, k2 = i_out 2^lgp + k)
            // fe_mul wants x tw < 0.9 R p: true as it stands while FB < 0.9 R / p (BN254: 31 of 169, BLS12-377: 7 of 438); BLS12-381 (63 of 70)
            // first brings x below 2p (one multiply-add chain, which also normalises it)
            constexpr bool REDUCE_FIRST = (long long)FB * 10 >= (long long)Fr::HEADROOM * 9;
            const u32 *row = A.ta + (((size_t)i2 << (lgp + 8)) + k + ((size_t)iq << lgp)) * 8;
            uint4 nlo = reinterpret_cast<const uint4 *>(row)[0], nhi = reinterpret_cast<const uint4 *>(row)[1];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const u32 w8[8] = {nlo.x, nlo.y, nlo.z, nlo.w, nhi.x, nhi.y, nhi.z, nhi.w};
                if (m + 1 < 8) {
                    const uint4 *nx = reinterpret_cast<const uint4 *>(row + ((size_t)(br3(m + 1) << 5) << lgp) * 8);
                    nlo = nx[0];
                    nhi = nx[1];
                }
                Fe<Fr> tw, x, v;
                fe_unpack(tw, w8);
                if constexpr (REDUCE_FIRST) {
                    x = e[m];
                    fe_reduce_mad_2p(x);
                } else
                    fe_norm(x, e[m]); // the last round leaves its outputs un-normalised (limbs < 3 * 2^30): a Montgomery product wants them below 2^30.5
                fe_mul(v, x, tw); // x < FB p, tw < p: below 0.9 R p; the result is tight and below 2p: it fits the 32-byte element and the next pass's input bound
                store_elem32(A.y + (base + ((size_t)(br3(m) << 5) << lgp)) * 8, v);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (A.cb == 0) { // one table: lgp == 0, k2 = i_out
            const unsigned row = i2 << A.ca;
            TwV<Fr> nxt;
            load_tw2(nxt, A.ta, row | iq);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const TwV<Fr> t = nxt;
                if (m + 1 < 8) load_tw2(nxt, A.ta, row | ((br3(m + 1) << 5) | iq));
                Fe<Fr> v;
                fe_mul_shoup<Fr, false>(v, e[m], t.w, t.q);
                store_elem32(A.y + (base + ((size_t)(br3(m) << 5) << lgp)) * 8, v);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else { // k2 = i_out 2^lgp + k: the low ca (<= lgp) bits are the thread's, the rest the element's
            TwV<Fr> ta;
            load_tw2(ta, A.ta, (i2 << A.ca) | (k & ((1u << A.ca) - 1)));
            const unsigned rowb = i2 << A.cb, khi0 = k >> A.ca, sh = lgp - A.ca;
            TwV<Fr> nxt;
            load_tw2(nxt, A.tb, rowb | ((iq << sh) | khi0));
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const TwV<Fr> t = nxt;
                if (m + 1 < 8) load_tw2(nxt, A.tb, rowb | ((((br3(m + 1) << 5) | iq) << sh) | khi0));
                Fe<Fr> v, u;
                fe_mul_shoup<Fr, false>(u, e[m], t.w, t.q);
                fe_mul_shoup<Fr, false>(v, u, ta.w, ta.q);
                store_elem32(A.y + (base + ((size_t)(br3(m) << 5) << lgp)) * 8, v);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
  } // tiles
}

// ---- a short LAST pass (radix 2, 4 or 8) entirely in registers ------------------------------------------------------------------------
// The last pass of a transform whose size is not a multiple of 8 bits (2^26 = 8 + 8 + 8 + 2), and the size-G transforms of the sharded
// transform's second step (G = 2, 4, 8 ranks): thread blk owns the 2^DEG elements x[blk + i S] of one sub-transform, runs its DEG
// rounds in registers (twiddles are powers of a 2^DEG-th root: scalar registers, 0 / 1 / 5 products per sub-transform), reduces to
// [0, p) and writes y[blk + i_out S] -- or, bit-reversed, the 2^DEG contiguous elements at bitrev(blk) 2^DEG.  No LDS, no
// barrier: a streaming kernel, 64 B of traffic per element.  Inputs below 3p; the twiddle between the passes was multiplied on by the
// pass before (k_ntt_pass8) or is not needed (second slab step).
struct SmallArgs {
    const u32 *x;
    u32 *y;
    const u32 *pq;      // max(1, 2^(DEG-1)) entries: powers of the 2^DEG-th root
    unsigned log_count; // log2 of the number of sub-transforms (= of the stride S)
    unsigned br_out;
};

template <class Fr, int DEG>
__global__ void __launch_bounds__(256) k_ntt_small(SmallArgs A)
{
    static_assert(DEG >= 1 && DEG <= 3, "radix 2, 4 or 8");
    constexpr int R = 1 << DEG;
    const size_t S = (size_t)1 << A.log_count;
    const size_t blk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= S) return;
    Fe<Fr> e[R];
#pragma unroll
    for (int i = 0; i < R; i++) load_elem32(e[i], A.x + (blk + i * S) * 8);
    // bounds 3 -> 7 -> 15 -> 31 (each round at most doubles and adds one): far below R / p for every supported field
    TwV<Fr> t;
    if constexpr (DEG == 1) {
        bfly<Fr, 3, false, false, true, 2>(e[0], e[1], nullptr, nullptr);
    } else if constexpr (DEG == 2) {
        load_tw2_uniform(t, A.pq, 1);
        bfly<Fr, 3, false, false, true, 0>(e[0], e[2], nullptr, nullptr);
        bfly<Fr, 3, true, false, true, 0>(e[1], e[3], t.w, t.q);
        bfly<Fr, 7, false, false, true, 2>(e[0], e[1], nullptr, nullptr);
        bfly<Fr, 7, false, false, true, 2>(e[2], e[3], nullptr, nullptr);
    } else {
        bfly<Fr, 3, false, false, true, 0>(e[0], e[4], nullptr, nullptr);
#pragma unroll
        for (int i = 1; i < 4; i++) {
            load_tw2_uniform(t, A.pq, i);
            bfly<Fr, 3, true, false, true, 0>(e[i], e[i + 4], t.w, t.q);
        }
        load_tw2_uniform(t, A.pq, 2);
        bfly<Fr, 7, false, false, true, 1>(e[0], e[2], nullptr, nullptr);
        bfly<Fr, 7, false, false, true, 1>(e[4], e[6], nullptr, nullptr);
        bfly<Fr, 7, true, false, true, 1>(e[1], e[3], t.w, t.q);
        bfly<Fr, 7, true, false, true, 1>(e[5], e[7], t.w, t.q);
#pragma unroll
        for (int i = 0; i < 8; i += 2) bfly<Fr, 15, false, false, true, 2>(e[i], e[i + 1], nullptr, nullptr);
    }
    // position i holds output bitrev(i)
#pragma unroll
    for (int i = 0; i < R; i++) {
        fe_reduce_mad_2p(e[i]);
        fe_reduce_once(e[i]);
        const unsigned i_out = (unsigned)(DEG == 1 ? i : (DEG == 2 ? (((i & 1) << 1) | (i >> 1)) : br3(i)));
        const size_t dst = A.br_out ? ((size_t)brev0((unsigned)blk, A.log_count) << DEG) + i : blk + i_out * S;
        store_elem32(A.y + dst * 8, e[i]);
    }
}

} // namespace panda_ntt8
