// Micro-benchmark: a Montgomery product of BN254 Fq built on the DOUBLE-PRECISION FMA pipe against the library's 9 x 29-bit v_mad_u64_u32
// product (fe_mul, fe29.h) -- the experiment behind DESIGN.md's "DFMA formulations lose" (VERDICT r3 item 7).
//
// Formulation (Emmart / Zheng / Weems, "Faster modular exponentiation using double precision floating point arithmetic on the GPU"):
//   * 5 limbs of 52 bits held as doubles (exact integers below 2^52), R = 2^260;
//   * a limb product a b < 2^104 is split by two fused multiply-adds under ROUND-TOWARD-ZERO:
//         hi = fma(a, b, 2^104)              = 2^104 + floor(a b / 2^52) 2^52        (the mantissa of hi IS the high half)
//         lo = fma(a, b, (2^104 + 2^52) - hi) = 2^52 + (a b mod 2^52)                 (exact: the high half cancels)
//   * the halves are accumulated per column as INTEGERS on the raw bit patterns (all terms of a kind share an exponent, whose
//     multiples are subtracted at the end), then carried, converted back and reduced word-serially (q_i = t_i p' mod 2^52).
// Count per product: 110 v_fma_f64 + 100 64-bit integer additions + the carries / conversions of 10 columns, against 162 v_mad_u64_u32
// + 9 v_mul_lo.  On gfx950 a 64-bit integer addition costs what a multiply-add costs (profiles/r01_ubench_int_rates.txt: add64 27.9,
// dfma 33.6, mad_u64_u32 28.45 T lane-ops/s), so the FMA route issues MORE quarter-rate slots per product, not fewer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../panda_amd/csrc ubench_dfma_mont.hip -o bin/ubench_dfma_mont
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "fe29.h"
using namespace panda29;
typedef Bn254Fq F;
typedef unsigned __int128 u128;

constexpr int NL = 5;
constexpr u64 M52 = (1ull << 52) - 1;

struct Consts {
    double p[NL];  // p in 52-bit limbs
    double pinv;   // -p^-1 mod 2^52
};

// ---- host reference: the same word-serial Montgomery product in integers -------------------------------------------------------
static void limbs52_of(u64 *out, const u32 *w8)
{
    u128 acc = 0;
    int bits = 0, k = 0;
    for (int i = 0; i < 8; i++) {
        acc |= (u128)w8[i] << bits;
        bits += 32;
        while (bits >= 52 && k < NL) {
            out[k++] = (u64)acc & M52;
            acc >>= 52;
            bits -= 52;
        }
    }
    while (k < NL) {
        out[k++] = (u64)acc & M52;
        acc >>= 52;
    }
}
static void host_montmul(u64 *r, const u64 *a, const u64 *b, const u64 *p, u64 pinv)
{
    u128 t[2 * NL + 1] = {0};
    for (int i = 0; i < NL; i++)
        for (int j = 0; j < NL; j++) t[i + j] += (u128)a[i] * b[j];
    for (int i = 0; i < NL; i++) {
        const u64 q = (u64)(((u128)((u64)t[i] & M52) * pinv) & M52);
        for (int j = 0; j < NL; j++) t[i + j] += (u128)q * p[j];
        t[i + 1] += t[i] >> 52;
    }
    for (int k = NL; k < 2 * NL; k++) {
        r[k - NL] = (u64)t[k] & M52;
        t[k + 1] += t[k] >> 52;
    }
}

// ---- device: the FMA formulation --------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 bits_of(double x) { return (u64)__double_as_longlong(x); }
__device__ __forceinline__ double dbl_of(u64 x) { return __longlong_as_double((long long)x); }
// integer below 2^52 -> the double with that value
__device__ __forceinline__ double to_double52(u64 v) { return dbl_of(v | (1075ull << 52)) - 4503599627370496.0; }

// The kernels switch the double-precision rounding mode to round-toward-zero with an inline-asm s_setreg: spelled as
// __builtin_amdgcn_s_setreg, hipcc's own mode tracking re-establishes round-to-nearest in front of the first v_fma_f64, and the device
// library's __ocml_fma_rtz_f64 is an emulation of ~15 instructions.
#define SET_ROUND_TOWARD_ZERO() asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory")

__device__ __forceinline__ void split_mac(u64 &col_lo, u64 &col_hi, double a, double b)
{
    const double C1 = 20282409603651670423947251286016.0;            // 2^104
    const double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0; // 2^104 + 2^52 (exact: ulp at 2^104 is 2^52)
    const double hi = __builtin_fma(a, b, C1);   // round toward zero (mode register, set by the kernel)
    const double lo = __builtin_fma(a, b, C2 - hi);
    col_hi += bits_of(hi);
    col_lo += bits_of(lo);
}

// r = a b / 2^260 mod p (values below 2p in, below 2p out; limbs as exact doubles)
__device__ __forceinline__ void dfma_montmul(double (&r)[NL], const double (&a)[NL], const double (&b)[NL], const Consts &c)
{
    constexpr u64 EXP_HI = (1023ull + 104) << 52, EXP_LO = (1023ull + 52) << 52;
    u64 col[2 * NL + 1];
    int nhi[2 * NL + 1], nlo[2 * NL + 1];
#pragma unroll
    for (int k = 0; k <= 2 * NL; k++) {
        col[k] = 0;
        nhi[k] = nlo[k] = 0;
    }
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) {
            split_mac(col[i + j], col[i + j + 1], a[i], b[j]);
            nlo[i + j]++;
            nhi[i + j + 1]++;
        }
#pragma unroll
    for (int i = 0; i < NL; i++) {
        // column i is complete: strip the exponent multiples, take its low word, q = t_i p' mod 2^52
        u64 t = col[i] - (u64)nhi[i] * EXP_HI - (u64)nlo[i] * EXP_LO;
        nhi[i] = nlo[i] = 0;
        const double ti = to_double52(t & M52);
        u64 ql = 0, qh = 0;
        split_mac(ql, qh, ti, c.pinv);
        const double q = to_double52((ql - EXP_LO) & M52);
        col[i] = t;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            split_mac(col[i + j], col[i + j + 1], q, c.p[j]);
            nlo[i + j]++;
            nhi[i + j + 1]++;
        }
        t = col[i] - (u64)nhi[i] * EXP_HI - (u64)nlo[i] * EXP_LO; // low 52 bits are zero now
        nhi[i] = nlo[i] = 0;
        col[i + 1] += t >> 52;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL; k++) {
        const u64 t = col[k] - (u64)nhi[k] * EXP_HI - (u64)nlo[k] * EXP_LO;
        r[k - NL] = to_double52(t & M52);
        col[k + 1] += t >> 52;
    }
}

#define ITERS 256
template <int VARIANT>
__global__ void __launch_bounds__(256) k_mul(u32 *out, const u32 *in, Consts c)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (VARIANT == 0) { // the library's product: two independent chains, as in ubench_mul.hip
        Fe<F> x, y, z, w;
        for (int j = 0; j < 9; j++) {
            x.l[j] = in[(i & 1023) * 18 + j] & LIMB_MASK;
            y.l[j] = in[(i & 1023) * 18 + 9 + j] & LIMB_MASK;
        }
        x.l[8] &= 0x3fffff;
        y.l[8] &= 0x3fffff;
        z = y;
        w = x;
        z.l[0] ^= 5;
        w.l[1] ^= 9;
        for (int it = 0; it < ITERS; it++) {
            fe_mul(x, x, y);
            fe_mul(z, z, w);
            fe_mul(y, y, x);
            fe_mul(w, w, z);
        }
        u32 s = 0;
        for (int j = 0; j < 9; j++) s += x.l[j] * 3 + y.l[j] * 5 + z.l[j] * 7 + w.l[j] * 11;
        out[i] = s;
    } else {
        SET_ROUND_TOWARD_ZERO();
        double x[NL], y[NL], z[NL], w[NL];
        for (int j = 0; j < NL; j++) {
            const u64 a = ((u64)in[(i & 1023) * 18 + 2 * j] | ((u64)in[(i & 1023) * 18 + 2 * j + 1] << 32)) & M52;
            const u64 b = ((u64)in[(i & 1023) * 18 + 2 * j + 8] | ((u64)in[(i & 1023) * 18 + 2 * j + 9] << 32)) & M52;
            x[j] = to_double52(j == NL - 1 ? (a & 0x3fffffffffull) : a); // top limb small: the value is below p
            y[j] = to_double52(j == NL - 1 ? (b & 0x3fffffffffull) : b);
        }
        for (int j = 0; j < NL; j++) {
            z[j] = y[j];
            w[j] = x[j];
        }
        for (int it = 0; it < ITERS; it++) {
            dfma_montmul(x, x, y, c);
            dfma_montmul(z, z, w, c);
            dfma_montmul(y, y, x, c);
            dfma_montmul(w, w, z, c);
        }
        u64 s = 0;
        for (int j = 0; j < NL; j++) s += bits_of(x[j]) * 3 + bits_of(y[j]) * 5 + bits_of(z[j]) * 7 + bits_of(w[j]) * 11;
        out[i] = (u32)s ^ (u32)(s >> 32);
    }
}

// one product per thread, limbs out: the correctness check against host_montmul
__global__ void k_check(u64 *out, const u64 *ab, Consts c, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SET_ROUND_TOWARD_ZERO();
    double a[NL], b[NL], r[NL];
    for (int j = 0; j < NL; j++) {
        a[j] = to_double52(ab[(size_t)i * 2 * NL + j]);
        b[j] = to_double52(ab[(size_t)i * 2 * NL + NL + j]);
    }
    dfma_montmul(r, a, b, c);
    for (int j = 0; j < NL; j++) out[(size_t)i * NL + j] = bits_of(r[j] + 4503599627370496.0) & M52;
}

template <int V>
static void run(const char *name, int blocks, u32 *out, u32 *in, const Consts &c)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k_mul<V><<<blocks, 256>>>(out, in, c);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k_mul<V><<<blocks, 256>>>(out, in, c);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double muls = (double)blocks * 256 * ITERS * 4;
    u32 chk;
    (void)hipMemcpy(&chk, out, 4, hipMemcpyDeviceToHost);
    printf("%-34s blocks=%5d  %8.3f ms  %8.2f G mulmod/s  (check %08x)\n", name, blocks, ms, muls / (ms * 1e-3) * 1e-9, chk);
}

int main()
{
    // p and -p^-1 mod 2^52 in 52-bit limbs
    u64 p52[NL];
    limbs52_of(p52, F::PW);
    u64 inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - p52[0] * inv; // Newton: p^-1 mod 2^64
    const u64 pinv = (0 - inv) & M52;
    Consts c;
    for (int j = 0; j < NL; j++) c.p[j] = (double)p52[j];
    c.pinv = (double)pinv;

    // correctness: 4096 random pairs below p against the integer reference
    const int n = 4096;
    std::vector<u64> ab((size_t)n * 2 * NL), want((size_t)n * NL), got((size_t)n * NL);
    u64 st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        st ^= st << 13;
        st ^= st >> 7;
        st ^= st << 17;
        return st;
    };
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < 2 * NL; j++) ab[(size_t)i * 2 * NL + j] = rnd() & M52;
        ab[(size_t)i * 2 * NL + NL - 1] &= p52[NL - 1] - 1; // top limbs below p's: values below p
        ab[(size_t)i * 2 * NL + 2 * NL - 1] &= p52[NL - 1] - 1;
        if (i < 4) // edge operands: 0, 1, p - 1 (limb-wise), all-ones low limbs
            for (int j = 0; j < NL; j++) ab[(size_t)i * 2 * NL + j] = i == 0 ? 0 : (i == 1 ? (j == 0) : (i == 2 ? p52[j] - (j == 0) : (j < NL - 1 ? M52 : 1)));
        host_montmul(&want[(size_t)i * NL], &ab[(size_t)i * 2 * NL], &ab[(size_t)i * 2 * NL + NL], p52, pinv);
    }
    u64 *d_ab, *d_out;
    (void)hipMalloc(&d_ab, ab.size() * 8);
    (void)hipMalloc(&d_out, got.size() * 8);
    (void)hipMemcpy(d_ab, ab.data(), ab.size() * 8, hipMemcpyHostToDevice);
    k_check<<<(n + 255) / 256, 256>>>(d_out, d_ab, c, n);
    (void)hipMemcpy(got.data(), d_out, got.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) bad += memcmp(&got[(size_t)i * NL], &want[(size_t)i * NL], NL * 8) != 0;
    printf("dfma_montmul vs the integer reference: %d of %d products differ\n", bad, n);

    u32 *out, *in;
    (void)hipMalloc(&out, 4 * 256 * 8192);
    (void)hipMalloc(&in, 1024 * 18 * 4);
    std::vector<u32> h(1024 * 18);
    for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2654435761u + 12345u);
    (void)hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int wps : {1, 2, 3, 4, 8}) {
        printf("--- %d waves/SIMD ---\n", wps);
        run<0>("montmul9x29 (v_mad_u64_u32)", 256 * wps, out, in, c);
        run<1>("montmul5x52 (v_fma_f64 hi/lo)", 256 * wps, out, in, c);
    }
    return bad != 0;
}
