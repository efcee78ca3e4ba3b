// ntt_radix9.h -- one radix-512 pass of the transform with the data held in registers (included by ntt.hip).
//
// Same construction as k_ntt_pass8 (ntt_radix8.h; the reference's pass protocol is src/cuda/core/unit/ntt/fft.cu:171-216), one bit wider:
// a workgroup of 256 threads owns a tile of FOUR 512-point sub-transforms (2048 elements), every thread owns eight elements and runs
// three register blocks of THREE butterfly rounds (distances 256/128/64, 32/16/8, 4/2/1) with two exchanges through LDS in between.
// It exists for the sizes whose bit count is not a multiple of eight: 2^26 runs as 9 + 9 + 8 instead of 8 + 8 + 8 + 2 (one pass over
// the data less, 13.1 instead of 14.3 products per element), 2^25 as 9 + 8 + 8, 2^27 as 9 + 9 + 9, 2^17 / 2^18 in two passes.
//
// What differs from the radix-256 pass:
//   * the middle block's twiddles depend on three index bits that are not register bits (i[2:0]); four waves carry two of them, so they
//     are no longer wave-uniform: block B reads them per lane from the LDS copy of the table, exactly as block A does (and the
//     unit-twiddle shortcut of wave 0 is gone: 12 products per thread in blocks A and B, 5 in block C -- the 8th roots of unity);
//   * the butterfly table has 256 entries (20 KB of LDS), so the exchange batches are PB = 3 limb planes (24 KB) to keep three
//     workgroups on a CU;
//   * the index maps.  Element (s, i), s < 4 the sub-transform and i < 512 the position, lives
//       in block A in thread (s, i0 = i[5:0]),       register m = i[8:6]   (lane = s | i0[3:0] << 2, wave = i0[5:4]: 128-byte runs from HBM)
//       in block B in thread (s, g = i[8:6], j = i[2:0]), register m = i[5:3]   (lane = s | g << 2 | j[0] << 5, wave = j[2:1])
//       in block C in thread (s, q = i[8:3]),        register m = i[2:0]   (lane = s | q[3:0] << 2, wave = q[5:4]; in a first pass that
//                                                                           is not the last: lane = bitrev6(q), wave = s, so that a store
//                                                                           instruction of a wave covers 2 KB of contiguous output)
//     and crosses LDS at word  s | (i[2:0] ^ i[8:6]) << 2 | i[5:0] << 5  (exchange 1) and
//     (i[8:6] ^ i[5:3]) | (s ^ i[5:4]) << 3 | i[5:0] << 5  (exchange 2): every store and load instruction of either side touches 32
//     distinct banks per half-wave.  tools/model_pass9.py runs these maps (and the pass formulas below) over a small field and counts the
//     bank multiplicities; tests/test_ntt_model.py keeps it in the CPU suite.
//
// Pass formulas (shared with k_ntt_pass8, radix 2^d, 2^lgp = product of the radices before): sub-transform blk = (blk_hi << lgp) | k reads
// x[blk + i (n >> d)] and writes output i_out to y[(blk_hi << (lgp + d)) | (i_out << lgp) | k], times W^(i2 k2) unless it is the last pass:
// W = w^(n >> (lgp + d + d2)) with 2^d2 the next radix, k2 = (i_out << lgp) | k, i2 = the top d2 bits of blk_hi.
#pragma once
#include "ntt_radix8.h"

namespace panda_ntt8 {

constexpr int SUBS9 = 4; // sub-transforms per tile

// bounds: as Plan8, nine rounds; rounds 0..5 multiply every difference, rounds 6..8 leave some (all) un-multiplied
template <class Fr, int B0>
struct Plan9 {
    int b[10];
    bool red[9];
    constexpr Plan9() : b{}, red{}
    {
        int B = B0;
        for (int r = 0; r < 9; r++) {
            b[r] = B;
            const int dplain = B + keff_of<Fr>(B);
            const int dmax = (r >= 6 && dplain > 3) ? dplain : 3;
            int next = 2 * B > dmax ? 2 * B : dmax;
            red[r] = r < 8 ? !round_ok<Fr>(next) : next >= (int)Fr::HEADROOM;
            if (red[r]) next = 3;
            B = next;
        }
        b[9] = B;
    }
};

template <class Fr, int B0>
inline constexpr Plan9<Fr, B0> plan9_v{};

template <class Fr, bool FIRST, bool LAST, int PB, int MINW>
__global__ void __launch_bounds__(THREADS, MINW) k_ntt_pass9(Pass8Args A)
{
    constexpr const Plan9<Fr, FIRST ? 2 : 3> &PL = plan9_v<Fr, FIRST ? 2 : 3>;
    __shared__ u32 s_x[PB * ELEMS];
    __shared__ __attribute__((aligned(16))) u32 s_tw[256 * TW2_STRIDE];

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned blk0 = blockIdx.x * SUBS9;
    const unsigned S = (1u << A.log_n) >> 9; // stride between the inputs of one sub-transform
    const unsigned lgp = A.lgp;

    // butterfly twiddles w_512^t, t < 256 -> LDS (entry tid: five 16-byte words per thread)
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(A.pq) + tid * 5;
        uint4 *l = reinterpret_cast<uint4 *>(s_tw) + tid * 5;
#pragma unroll
        for (int j = 0; j < 5; j++) l[j] = g[j];
    }

    // ---- block A: thread (s, i0) holds i = i0 + 64 m; rounds 0..2 (distances 256, 128, 64)
    Fe<Fr> e[8];
    unsigned s = lane & 3;
    const unsigned i0 = (lane >> 2) | (wave << 4);
    {
        const size_t base = (size_t)(blk0 + s) + (size_t)i0 * S, step = (size_t)64 * S;
#pragma unroll
        for (int m = 0; m < 8; m++) load_elem32(e[m], A.x + (base + m * step) * 8);
    }
    __syncthreads(); // s_tw complete
    {
        TwV<Fr> t;
#pragma unroll
        for (int m = 0; m < 4; m++) { // round 0: pairs (m, m + 4), twiddle index i mod 256
            load_tw2(t, s_tw, i0 + 64 * m);
            bfly<Fr, PL.b[0], true, PL.red[0], false, 0>(e[m], e[m + 4], t.w, t.q);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) { // round 1: pairs (m, m + 2), twiddle index 2 (i mod 128)
            load_tw2(t, s_tw, 2 * (i0 + 64 * h));
            bfly<Fr, PL.b[1], true, PL.red[1], false, 1>(e[h], e[h + 2], t.w, t.q);
            bfly<Fr, PL.b[1], true, PL.red[1], false, 1>(e[h + 4], e[h + 6], t.w, t.q);
        }
        load_tw2(t, s_tw, 4 * i0); // round 2: pairs (m, m + 1), twiddle index 4 (i mod 64)
#pragma unroll
        for (int m = 0; m < 8; m += 2) bfly<Fr, PL.b[2], true, PL.red[2], false, 0>(e[m], e[m + 1], t.w, t.q);
    }

    // ---- exchange 1, then block B: thread (s, g, j) holds i = 64 g + 8 m + j; rounds 3..5 (distances 32, 16, 8)
    const unsigned g = (lane >> 2) & 7, j = (lane >> 5) | (wave << 1);
    {
        unsigned wa[8];
        const unsigned wbase = s | ((i0 & 7) << 2) | (i0 << 5);
#pragma unroll
        for (int m = 0; m < 8; m++) wa[m] = wbase ^ (m << 2);
        const unsigned ra = s | ((j ^ g) << 2) | (j << 5);
        exchange<Fr, PB, 256>(e, s_x, wa, ra);
    }
    {
        TwV<Fr> t;
#pragma unroll
        for (int m = 0; m < 4; m++) { // round 3: twiddle index 8 (i mod 32)
            load_tw2(t, s_tw, 8 * (8 * m + j));
            bfly<Fr, PL.b[3], true, PL.red[3], false, 1>(e[m], e[m + 4], t.w, t.q);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) { // round 4: twiddle index 16 (i mod 16)
            load_tw2(t, s_tw, 16 * (8 * h + j));
            bfly<Fr, PL.b[4], true, PL.red[4], false, 0>(e[h], e[h + 2], t.w, t.q);
            bfly<Fr, PL.b[4], true, PL.red[4], false, 0>(e[h + 4], e[h + 6], t.w, t.q);
        }
        load_tw2(t, s_tw, 32 * j); // round 5: twiddle index 32 (i mod 8)
#pragma unroll
        for (int m = 0; m < 8; m += 2) bfly<Fr, PL.b[5], true, PL.red[5], false, 1>(e[m], e[m + 1], t.w, t.q);
    }

    // ---- exchange 2, then block C: thread (s, q) holds i = 8 q + m; rounds 6..8 (distances 4, 2, 1)
    unsigned q;
    {
        __syncthreads(); // everyone has read exchange 1's last batch
        unsigned wa[8];
#pragma unroll
        for (int m = 0; m < 8; m++) wa[m] = (g ^ m) | ((s ^ (m >> 1)) << 3) | (j << 5) | (m << 8);
        if (!LAST && lgp == 0) { // a sub-transform's outputs are contiguous in i_out = bitrev9(i): a wave per sub-transform, lane = bitrev6(q)
            s = wave;
            q = brev(lane, 6);
        } else { // consecutive sub-transforms are contiguous
            q = (lane >> 2) | (wave << 4);
        }
        const unsigned ra = ((q >> 3) ^ (q & 7)) | ((s ^ ((q >> 1) & 3)) << 3) | ((q & 7) << 8);
        exchange<Fr, PB, 32>(e, s_x, wa, ra);
    }
    {
        TwV<Fr> t;
        // round 6: twiddle w^(64 (i mod 4)) -- 1 and the three other 8th roots of unity in the first half
        bfly<Fr, PL.b[6], false, PL.red[6], true, 0>(e[0], e[4], nullptr, nullptr);
#pragma unroll
        for (int m = 1; m < 4; m++) {
            load_tw2_uniform(t, A.pq, 64 * m);
            bfly<Fr, PL.b[6], true, PL.red[6], true, 0>(e[m], e[m + 4], t.w, t.q);
        }
        // round 7: 1 for even i, the 4th root of unity for odd i
        load_tw2_uniform(t, A.pq, 128);
        bfly<Fr, PL.b[7], false, PL.red[7], true, 1>(e[0], e[2], nullptr, nullptr);
        bfly<Fr, PL.b[7], false, PL.red[7], true, 1>(e[4], e[6], nullptr, nullptr);
        bfly<Fr, PL.b[7], true, PL.red[7], true, 1>(e[1], e[3], t.w, t.q);
        bfly<Fr, PL.b[7], true, PL.red[7], true, 1>(e[5], e[7], t.w, t.q);
        // round 8: 1
#pragma unroll
        for (int m = 0; m < 8; m += 2) bfly<Fr, PL.b[8], false, PL.red[8], true, PL.red[8] ? 1 : 2>(e[m], e[m + 1], nullptr, nullptr);
    }
    constexpr int FB = PL.b[9];
    static_assert(FB < (int)Fr::HEADROOM && FB < 512, "final bound");
    static_assert(LAST || ((unsigned long long)(Fr::HEADROOM + FB + 1) * ((unsigned long long)Fr::PW[Fr::L - 1] + 1) < (unsigned long long)Fr::HEADROOM << 32),
                  "an inter-pass output product would not fit the 32-byte element");

    // ---- output: register m holds output i_out = 64 bitrev3(m) + bitrev6(q) of sub-transform blk = blk0 + s
    const unsigned blk = blk0 + s;
    const unsigned k = blk & ((1u << lgp) - 1);
    const unsigned iq = brev(q, 6);
    const size_t base = ((size_t)(blk - k) << 9) + k + ((size_t)iq << lgp);
    if constexpr (LAST) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            fe_reduce_mad_2p(e[m]);
            fe_reduce_once(e[m]);
            store_elem32(A.y + (base + ((size_t)(br3(m) << 6) << lgp)) * 8, e[m]);
        }
    } else {
        const unsigned i2 = (blk >> lgp) >> A.i2_shift;
        if (A.wide) { // one streamed table over the whole index range, one Montgomery product (see k_ntt_pass8)
            constexpr bool REDUCE_FIRST = (long long)FB * 10 >= (long long)Fr::HEADROOM * 9;
            const u32 *row = A.ta + (((size_t)i2 << (lgp + 9)) + k + ((size_t)iq << lgp)) * 8;
            uint4 nlo = reinterpret_cast<const uint4 *>(row)[0], nhi = reinterpret_cast<const uint4 *>(row)[1];
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const u32 w8[8] = {nlo.x, nlo.y, nlo.z, nlo.w, nhi.x, nhi.y, nhi.z, nhi.w};
                if (m + 1 < 8) {
                    const uint4 *nx = reinterpret_cast<const uint4 *>(row + ((size_t)(br3(m + 1) << 6) << lgp) * 8);
                    nlo = nx[0];
                    nhi = nx[1];
                }
                Fe<Fr> tw, x, v;
                fe_unpack(tw, w8);
                if constexpr (REDUCE_FIRST) {
                    x = e[m];
                    fe_reduce_mad_2p(x);
                } else
                    fe_norm(x, e[m]);
                fe_mul(v, x, tw);
                store_elem32(A.y + (base + ((size_t)(br3(m) << 6) << lgp)) * 8, v);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (A.cb == 0) { // one table: lgp == 0, k2 = i_out
            const unsigned row = i2 << A.ca;
            TwV<Fr> nxt;
            load_tw2(nxt, A.ta, row | iq);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const TwV<Fr> t = nxt;
                if (m + 1 < 8) load_tw2(nxt, A.ta, row | ((br3(m + 1) << 6) | iq));
                Fe<Fr> v;
                fe_mul_shoup<Fr, false>(v, e[m], t.w, t.q);
                store_elem32(A.y + (base + ((size_t)(br3(m) << 6) << lgp)) * 8, v);
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
                if (m + 1 < 8) load_tw2(nxt, A.tb, rowb | ((((br3(m + 1) << 6) | iq) << sh) | khi0));
                Fe<Fr> v, u;
                fe_mul_shoup<Fr, false>(u, e[m], t.w, t.q);
                fe_mul_shoup<Fr, false>(v, u, ta.w, ta.q);
                store_elem32(A.y + (base + ((size_t)(br3(m) << 6) << lgp)) * 8, v);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

} // namespace panda_ntt8
