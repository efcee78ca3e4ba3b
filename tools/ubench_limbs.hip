// Micro-benchmark: Montgomery products over BLS12-377 Fq (377 bits) on
//   A   14 limbs of 29 bits, one 64-bit column accumulator        -- the library's layout: 196 + 196 multiply-adds, columns < 28 * 2^58
//   A0  the same, without the 14 products by the modulus' top limb, which is zero (377 = 13 * 29)  -- what fe29.h does from round 6 on
//   B   13 limbs of 30 bits: 169 + 169 multiply-adds.  26 products of < 2^60 do not fit a 64-bit column, so the a*b products and the
//       m*p products of a column are accumulated SEPARATELY and their carries are kept apart:
//           lows   = lo32(accA) + lo32(accM)                      m_k = (-lows) mod 2^30   (p = 1 mod 2^46: -p^-1 = -1, and P[0] = 1)
//           accM  += m_k                                          now accA + accM = 0 mod 2^30
//           carryA = accA >> 30;   carryM = (accM >> 30) + (lo30(accA) != 0)
//       i.e. per column one more 64-bit shift, an add, a compare and a 64-bit add than layout A.
// Both in the same dependent-chain workload (two independent chains of products, as inside a mixed addition), at 2 / 3 / 4 waves per
// SIMD; every variant's outputs are checked against big-integer arithmetic on the host (x * y = out * R mod p).
//   hipcc --offload-arch=gfx950 -O3 -I../panda_amd/csrc ubench_limbs.hip -o bin/ubench_limbs
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "fe29_chain.h"
using panda29::MacChain;
typedef uint32_t u32;
typedef uint64_t u64;

struct L29 {
    static constexpr int N = 14, B = 29;
    static constexpr u32 P[14] = {0x1, 0x8460000, 0x21, 0x16ba8860, 0x14800170, 0x1117dd04, 0xe3c7bcd, 0x1e601ea2, 0x1b1a22d9, 0x3650a49, 0x118ec170, 0xf8a21d5, 0x1ae3a461, 0x0};
};
struct L30 {
    static constexpr int N = 13, B = 30;
    static constexpr u32 P[13] = {0x1, 0x14230000, 0x8, 0x2d7510c, 0x9480017, 0xd88bee8, 0x1138f1ef, 0x367cc03d, 0x93b1a22, 0x1701b285, 0xeac63b0, 0x1185f144, 0x1ae3a};
};

// ---- layout A / A0: one accumulator (the structure of fe29_mul_col / fe29_reduce_col)
template <class F, bool SKIP0, int K>
__device__ __forceinline__ void colA(u64 &acc, const u32 *a, const u32 *b, u32 *m, u32 *out)
{
    constexpr int N = F::N;
    constexpr u32 MASK = (1u << F::B) - 1;
    if constexpr (K == 0)
        MacChain<1>::vv0(acc, a, b);
    else if constexpr (K < N)
        MacChain<K + 1>::vv(acc, a, b + K);
    else
        MacChain<2 * N - 1 - K>::vv(acc, a + (K - N + 1), b + (N - 1));
    if constexpr (K < N) {
        if constexpr (SKIP0 && K == N - 1)
            MacChain<K - 1>::vs(acc, m + 1, &F::P[K - 1]);
        else
            MacChain<K>::vs(acc, m, &F::P[K]);
        m[K] = (0u - (u32)acc) & MASK; // -p^-1 = -1 mod 2^B
        MacChain<1>::vs(acc, m + K, &F::P[0]);
    } else {
        if constexpr (SKIP0)
            MacChain<2 * N - 2 - K>::vs(acc, m + (K - N + 2), &F::P[N - 2]);
        else
            MacChain<2 * N - 1 - K>::vs(acc, m + (K - N + 1), &F::P[N - 1]);
        out[K - N] = (u32)acc & MASK;
    }
    acc >>= F::B;
}
template <class F, bool SKIP0, int... K>
__device__ __forceinline__ void colsA(u64 &acc, const u32 *a, const u32 *b, u32 *m, u32 *out, std::integer_sequence<int, K...>)
{
    (colA<F, SKIP0, K>(acc, a, b, m, out), ...);
}
template <class F, bool SKIP0>
__device__ __forceinline__ void mulA(u32 *r, const u32 *a, const u32 *b)
{
    constexpr int N = F::N;
    u32 m[N], out[N];
    u64 acc = 0;
    colsA<F, SKIP0>(acc, a, b, m, out, std::make_integer_sequence<int, 2 * N - 1>());
    out[N - 1] = (u32)acc;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = out[i];
}

// ---- layout B: 13 x 30 bits, the a*b and the m*p products of a column in accumulators of their own
template <class F, int K>
__device__ __forceinline__ void colB(u64 &accA, u64 &accM, const u32 *a, const u32 *b, u32 *m, u32 *out)
{
    constexpr int N = F::N;
    constexpr u32 MASK = (1u << F::B) - 1;
    if constexpr (K < N)
        MacChain<K + 1>::vv(accA, a, b + K);
    else
        MacChain<2 * N - 1 - K>::vv(accA, a + (K - N + 1), b + (N - 1));
    if constexpr (K < N) {
        MacChain<K>::vs(accM, m, &F::P[K]);
        const u32 lows = (u32)accA + (u32)accM;
        m[K] = (0u - lows) & MASK;
        accM += m[K]; // m_k * P[0], P[0] = 1
        const u32 c = ((u32)accA & MASK) != 0 ? 1u : 0u; // accA + accM = 0 mod 2^30: their low parts add up to 0 or to 2^30
        accA >>= F::B;
        accM = (accM >> F::B) + c;
    } else {
        MacChain<2 * N - 1 - K>::vs(accM, m + (K - N + 1), &F::P[N - 1]);
        const u32 lows = ((u32)accA & MASK) + ((u32)accM & MASK);
        out[K - N] = lows & MASK;
        accA >>= F::B;
        accM = (accM >> F::B) + (lows >> F::B);
    }
}
template <class F, int... K>
__device__ __forceinline__ void colsB(u64 &accA, u64 &accM, const u32 *a, const u32 *b, u32 *m, u32 *out, std::integer_sequence<int, K...>)
{
    (colB<F, K>(accA, accM, a, b, m, out), ...);
}
template <class F>
__device__ __forceinline__ void mulB(u32 *r, const u32 *a, const u32 *b)
{
    constexpr int N = F::N;
    u32 m[N], out[N];
    u64 accA = 0, accM = 0;
    colsB<F>(accA, accM, a, b, m, out, std::make_integer_sequence<int, 2 * N - 1>());
    out[N - 1] = (u32)(accA + accM);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = out[i];
}

#define ITERS 256
template <class F, int VARIANT, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_mul(u32 *out, const u32 *in, int iters)
{
    constexpr int N = F::N;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 x[N], y[N], z[N], w[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        x[j] = in[(i & 1023) * 2 * N + j];
        y[j] = in[(i & 1023) * 2 * N + N + j];
        z[j] = y[j];
        w[j] = x[j];
    }
    z[0] ^= 5;
    w[1] ^= 9;
#pragma unroll 1
    for (int it = 0; it < iters; it++) { // two independent chains, like the independent products inside a mixed addition
        if constexpr (VARIANT == 2) {
            mulB<F>(x, x, y);
            mulB<F>(z, z, w);
            mulB<F>(y, y, x);
            mulB<F>(w, w, z);
        } else {
            mulA<F, VARIANT == 1>(x, x, y);
            mulA<F, VARIANT == 1>(z, z, w);
            mulA<F, VARIANT == 1>(y, y, x);
            mulA<F, VARIANT == 1>(w, w, z);
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        out[(size_t)i * 4 * N + j] = x[j];
        out[(size_t)i * 4 * N + N + j] = y[j];
        out[(size_t)i * 4 * N + 2 * N + j] = z[j];
        out[(size_t)i * 4 * N + 3 * N + j] = w[j];
    }
}

// ---- host big integers (448 bits in 7 x u64), only for the check
struct Big {
    u64 w[7];
};
static const Big PBIG = {{0x8508c00000000001ull, 0x170b5d4430000000ull, 0x1ef3622fba094800ull, 0x1a22d9f300f5138full, 0xc63b05c06ca1493bull, 0x01ae3a4617c510eaull, 0}};
static int cmp(const Big &a, const Big &b)
{
    for (int i = 6; i >= 0; i--)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
    return 0;
}
static void sub(Big &a, const Big &b)
{
    unsigned __int128 br = 0;
    for (int i = 0; i < 7; i++) {
        unsigned __int128 d = (unsigned __int128)a.w[i] - b.w[i] - br;
        a.w[i] = (u64)d;
        br = (d >> 64) & 1;
    }
}
static void add(Big &a, const Big &b)
{
    unsigned __int128 c = 0;
    for (int i = 0; i < 7; i++) {
        c += (unsigned __int128)a.w[i] + b.w[i];
        a.w[i] = (u64)c;
        c >>= 64;
    }
}
static void reduce(Big &a)
{
    while (cmp(a, PBIG) >= 0) sub(a, PBIG);
}
static Big from_limbs(const u32 *l, int n, int bits)
{
    Big r;
    memset(&r, 0, sizeof r);
    for (int i = n - 1; i >= 0; i--) { // r = r * 2^bits + l[i], values stay below 2^448
        for (int s = 0; s < bits; s++) {
            for (int k = 6; k > 0; k--) r.w[k] = (r.w[k] << 1) | (r.w[k - 1] >> 63);
            r.w[0] <<= 1;
        }
        Big t;
        memset(&t, 0, sizeof t);
        t.w[0] = l[i];
        add(r, t);
    }
    return r;
}
static Big dbl_mod(Big a)
{
    Big b = a;
    add(a, b);
    reduce(a);
    return a;
}
static Big mul_mod(Big a, Big b) // a, b < p
{
    Big r;
    memset(&r, 0, sizeof r);
    for (int bit = 383; bit >= 0; bit--) {
        r = dbl_mod(r);
        if ((b.w[bit / 64] >> (bit % 64)) & 1) {
            add(r, a);
            reduce(r);
        }
    }
    return r;
}

template <class F, int VARIANT, int WAVES>
static bool run(const char *name, int blocks, u32 *d_out, u32 *d_in, const std::vector<u32> &h_in, bool check)
{
    constexpr int N = F::N;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    bool ok = true;
    if (check) { // ONE round of the four products on a few lanes, against integers: out * R = x * y mod p
        k_mul<F, VARIANT, WAVES><<<4, 256>>>(d_out, d_in, 1);
        hipDeviceSynchronize();
        std::vector<u32> o(1024 * 4 * N);
        hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
        Big R;
        memset(&R, 0, sizeof R);
        R.w[0] = 1;
        for (int s = 0; s < N * F::B; s++) R = dbl_mod(R);
        for (int lane = 0; lane < 1024; lane += 37) {
            Big x = from_limbs(&h_in[lane * 2 * N], N, F::B), y = from_limbs(&h_in[lane * 2 * N + N], N, F::B);
            reduce(x);
            reduce(y);
            Big x1 = from_limbs(&o[(size_t)lane * 4 * N], N, F::B); // x1 = x * y / R
            reduce(x1);
            Big lhs = mul_mod(x1, R), rhs = mul_mod(x, y);
            if (cmp(lhs, rhs) != 0) ok = false;
            Big y1 = from_limbs(&o[(size_t)lane * 4 * N + N], N, F::B); // y1 = y * x1 / R
            reduce(y1);
            lhs = mul_mod(y1, R);
            rhs = mul_mod(y, x1);
            if (cmp(lhs, rhs) != 0) ok = false;
            // outputs must stay chainable: below 2^(B N) with every limb but the top inside its radix
            for (int j = 0; j + 1 < N; j++)
                if (o[(size_t)lane * 4 * N + j] >> F::B) ok = false;
        }
    }
    k_mul<F, VARIANT, WAVES><<<blocks, 256>>>(d_out, d_in, ITERS);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k_mul<F, VARIANT, WAVES><<<blocks, 256>>>(d_out, d_in, ITERS);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double muls = (double)blocks * 256 * ITERS * 4;
    printf("%-34s %d waves/SIMD  %8.3f ms  %8.2f G products/s%s\n", name, WAVES, ms, muls / (ms * 1e-3) * 1e-9, check ? (ok ? "   checked against integers: ok" : "   CHECK FAILED") : "");
    return ok;
}

int main()
{
    u32 *out, *in;
    hipMalloc(&out, (size_t)4 * 256 * 1024 * 4 * 14 * 4);
    hipMalloc(&in, 1024 * 28 * 4);
    bool ok = true;
    for (int layout = 0; layout < 2; layout++) {
        const int N = layout ? 13 : 14, B = layout ? 30 : 29;
        std::vector<u32> h(1024 * 2 * N);
        for (size_t i = 0; i < h.size(); i++) {
            h[i] = (u32)(i * 2654435761u + 12345u) & ((1u << B) - 1);
            if (i % N == (size_t)N - 1) h[i] &= layout ? 0x1ffffu : 0x1u; // values below 2^378 (< 2p + ...): tight operands
        }
        hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        if (!layout) {
            ok &= run<L29, 0, 2>("A  14 x 29, one accumulator", 256 * 2, out, in, h, true);
            run<L29, 0, 3>("A  14 x 29, one accumulator", 256 * 3, out, in, h, false);
            run<L29, 0, 4>("A  14 x 29, one accumulator", 256 * 4, out, in, h, false);
            ok &= run<L29, 1, 2>("A0 14 x 29, zero top limb skipped", 256 * 2, out, in, h, true);
            run<L29, 1, 3>("A0 14 x 29, zero top limb skipped", 256 * 3, out, in, h, false);
            run<L29, 1, 4>("A0 14 x 29, zero top limb skipped", 256 * 4, out, in, h, false);
        } else {
            ok &= run<L30, 2, 2>("B  13 x 30, split accumulators", 256 * 2, out, in, h, true);
            run<L30, 2, 3>("B  13 x 30, split accumulators", 256 * 3, out, in, h, false);
            run<L30, 2, 4>("B  13 x 30, split accumulators", 256 * 4, out, in, h, false);
        }
    }
    return ok ? 0 : 1;
}
