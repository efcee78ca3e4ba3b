// debug_gen.hip -- element-wise diagnostics kernels and the on-device synthetic input generators.
//
// panda_debug_field_op / panda_debug_curve_op run the very device functions the MSM and NTT kernels are
// built from (fe29.h, curve29.h) over arrays, so that the parity tests can compare them one element at a
// time with the oracle (reference counterparts: field.cuh:139-619, projective.cuh:163-314).
// panda_gen_scalars / panda_gen_bases build the SURVEY section 8d inputs directly in HBM: 2^26 bases
// (4 GiB) cannot sensibly be generated on the host and shipped over PCIe.
#include "curve29_quad.h"
#include "panda_internal.h"

using namespace panda29;

namespace {

// Out-of-line copies of the heavy group operations for the 14-limb fields: these diagnostic kernels are not timed, and
// inlining every operation for every curve made this file the longest compile of the library.
template <class F>
__device__ __noinline__ void to_affine_outlined(Fe<F> &x, Fe<F> &y, const Xyzz<F> &p)
{
    xyzz_to_affine_internal(x, y, p);
}
template <class F>
__device__ __noinline__ void madd_outlined(Xyzz<F> &acc, const Fe<F> &x, const Fe<F> &y, bool inf)
{
    xyzz_madd(acc, x, y, inf);
}
template <class F>
__device__ __noinline__ void add_outlined(Xyzz<F> &acc, const Xyzz<F> &q)
{
    xyzz_add(acc, q);
}
template <class F>
__device__ __noinline__ void dbl_outlined(Xyzz<F> &r, const Xyzz<F> &p)
{
    xyzz_dbl(r, p);
}


template <class F>
__global__ void __launch_bounds__(256) k_field_op(unsigned op, u32 *__restrict__ r, const u32 *__restrict__ a, const u32 *__restrict__ b, size_t n)
{
    constexpr int L = F::L;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 wa[L], wb[L], wr[L];
#pragma unroll
    for (int k = 0; k < L; k++) {
        wa[k] = a[i * L + k];
        wb[k] = b ? b[i * L + k] : 0;
    }
    Fe<F> x, y, z;
    if (op == 4) { // canonical -> Montgomery wire
        Fe<F> t, k;
        fe_unpack(t, wa);
        fe_const(k, F::K_TOINT);
        fe_mul(x, t, k);
        fe_to_wire(wr, x);
    } else if (op == 5) { // Montgomery wire -> canonical
        fe_wire_to_canonical<F>(wr, wa);
    } else {
        fe_from_wire(x, wa);
        fe_from_wire(y, wb);
        if (op == 0) fe_add(z, x, y);
        else if (op == 1) fe_sub<F, 2>(z, x, y);
        else if (op == 2) fe_mul(z, x, y);
        else if (op == 3) fe_sqr(z, x);
        else fe_inv(z, x); // a^(p-2): the inversion k_table_step and the host-side output conversion use; 0 -> 0
        fe_to_wire(wr, z);
    }
#pragma unroll
    for (int k = 0; k < L; k++) r[i * L + k] = wr[k];
}

template <class F>
__global__ void __launch_bounds__(128) k_curve_op(unsigned op, u32 *__restrict__ r, const u32 *__restrict__ a, const u32 *__restrict__ b, size_t n)
{
    constexpr int L = F::L;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 wa[3 * L], wb[3 * L], wr[3 * L];
#pragma unroll
    for (int k = 0; k < 3 * L; k++) wa[k] = a[i * 3 * L + k];
    Xyzz<F> p, q;
    xyzz_from_jacobian_wire(p, wa);
    if (op == 0) {
#pragma unroll
        for (int k = 0; k < 2 * L; k++) wb[k] = b[i * 2 * L + k];
        Fe<F> x, y;
        bool inf = affine_from_wire(x, y, wb);
        madd_outlined(p, x, y, inf);
    } else if (op == 1) {
#pragma unroll
        for (int k = 0; k < 3 * L; k++) wb[k] = b[i * 3 * L + k];
        xyzz_from_jacobian_wire(q, wb);
        add_outlined(p, q);
    } else {
        dbl_outlined(q, p);
        p = q;
    }
    xyzz_to_jacobian_wire(wr, p);
#pragma unroll
    for (int k = 0; k < 3 * L; k++) r[i * 3 * L + k] = wr[k];
}

// ops 3 / 4: the four-lane addition / doubling of curve29_quad.h (what the MSM's fix-up and bucket-reduction trees run), one QUAD per element
template <class F>
__device__ __noinline__ void add_quad_outlined(Xyzz<F> &acc, const Xyzz<F> &q, unsigned role)
{
    xyzz_add_quad(acc, q, role);
}
template <class F>
__device__ __noinline__ void dbl_quad_outlined(Xyzz<F> &r, const Xyzz<F> &p, unsigned role)
{
    xyzz_dbl_quad(r, p, role);
}

template <class F>
__global__ void __launch_bounds__(128) k_curve_op_quad(unsigned op, u32 *__restrict__ r, const u32 *__restrict__ a, const u32 *__restrict__ b, size_t n)
{
    constexpr int L = F::L;
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const unsigned role = threadIdx.x & 3u;
    if (i >= n) return; // whole quads leave together
    u32 wa[3 * L], wb[3 * L], wr[3 * L];
#pragma unroll
    for (int k = 0; k < 3 * L; k++) wa[k] = a[i * 3 * L + k];
    Xyzz<F> p, q;
    xyzz_from_jacobian_wire(p, wa);
    if (op == 3) {
#pragma unroll
        for (int k = 0; k < 3 * L; k++) wb[k] = b[i * 3 * L + k];
        xyzz_from_jacobian_wire(q, wb);
        add_quad_outlined(p, q, role);
    } else {
        dbl_quad_outlined(q, p, role);
        p = q;
    }
    xyzz_to_jacobian_wire(wr, p);
    if (role == (unsigned)(i & 3)) { // a different lane of the quad answers for each element: all four must hold the result
#pragma unroll
        for (int k = 0; k < 3 * L; k++) r[i * 3 * L + k] = wr[k];
    }
}

// ---- generators: same functions of (seed, index) as oracle/gen.c
__host__ __device__ __forceinline__ u64 splitmix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <class F>
__global__ void __launch_bounds__(256) k_gen_scalars(u64 seed, u64 first, u64 n, u32 *__restrict__ out)
{
    constexpr int L = F::L;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr unsigned top_bits = F::BITS - 32 * (L - 1);
    constexpr u32 top_mask = top_bits >= 32 ? 0xffffffffu : ((1u << top_bits) - 1);
    u64 base = seed ^ ((first + i) * 0xD1342543DE82EF95ull);
    u32 v[L];
    for (unsigned attempt = 0;; ) {
#pragma unroll
        for (int k = 0; k < L / 2; k++) {
            u64 w = splitmix64(base + (u64)attempt * (L / 2) + k);
            v[2 * k] = (u32)w;
            v[2 * k + 1] = (u32)(w >> 32);
        }
        v[L - 1] &= top_mask;
        bool lt = false, decided = false;
#pragma unroll
        for (int k = L - 1; k >= 0; k--) {
            if (!decided && v[k] != F::PW[k]) {
                lt = v[k] < F::PW[k];
                decided = true;
            }
        }
        if (lt) break;
        if (++attempt == 64) {
            v[L - 1] &= top_mask >> 1;
            break;
        }
    }
#pragma unroll
    for (int k = 0; k < L; k++) out[i * L + k] = v[k];
}

__host__ __device__ __forceinline__ u64 gen_multiplier(u64 seed, u64 i) { return splitmix64(seed ^ (0x9E3779B97F4A7C15ull * (i + 1))) | 1ull; }

// table[j][d-1] = d * 256^j * G in internal affine form (2*N limbs each), built by one thread per j
template <class F>
__global__ void k_gen_table(u32 *__restrict__ table, const u32 *__restrict__ gen_wire)
{
    constexpr int N = F::N, L = F::L;
    unsigned j = threadIdx.x;
    if (j >= 8) return;
    u32 gw[2 * L];
#pragma unroll
    for (int k = 0; k < 2 * L; k++) gw[k] = gen_wire[k];
    Fe<F> bx, by;
    affine_from_wire(bx, by, gw);
    // base = 256^j * G: 8*j doublings of G, then normalise to affine
    Xyzz<F> acc, d;
    xyzz_from_affine(acc, bx, by);
    for (unsigned k = 0; k < 8 * j; k++) {
        dbl_outlined(d, acc);
        acc = d;
    }
    Fe<F> ax, ay;
    to_affine_outlined(ax, ay, acc);
    bx = ax;
    by = ay;
    xyzz_set_identity(acc);
    for (unsigned dgt = 1; dgt <= 255; dgt++) {
        madd_outlined(acc, bx, by, false);
        to_affine_outlined(ax, ay, acc);
        u32 *dst = table + ((size_t)j * 255 + (dgt - 1)) * 2 * N;
        for (int k = 0; k < N; k++) {
            dst[k] = ax.l[k];
            dst[N + k] = ay.l[k];
        }
    }
}

template <class F>
__global__ void __launch_bounds__(128) k_gen_bases(u64 seed, u64 first, u64 n, const u32 *__restrict__ table, u32 *__restrict__ out)
{
    constexpr int N = F::N, L = F::L;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 m = gen_multiplier(seed, first + i);
    Xyzz<F> acc;
    xyzz_set_identity(acc);
    for (unsigned j = 0; j < 8; j++) {
        unsigned d = (unsigned)(m >> (8 * j)) & 255u;
        if (!d) continue;
        const u32 *src = table + ((size_t)j * 255 + (d - 1)) * 2 * N;
        Fe<F> x, y;
#pragma unroll
        for (int k = 0; k < N; k++) {
            x.l[k] = src[k];
            y.l[k] = src[N + k];
        }
        madd_outlined(acc, x, y, false);
    }
    Fe<F> ax, ay;
    u32 w[2 * L];
    to_affine_outlined(ax, ay, acc);
    fe_to_wire(w, ax);
    fe_to_wire(w + L, ay);
#pragma unroll
    for (int k = 0; k < 2 * L; k++) out[i * 2 * L + k] = w[k];
}

// generators G1 in wire form (canonical coordinates converted on the host with fe29)
template <class F>
void generator_wire(u32 *out, unsigned curve)
{
    constexpr int L = F::L;
    u32 cx[L] = {0}, cy[L] = {0};
    if (curve == 0) {
        cx[0] = 1;
        cy[0] = 2;
    } else if (curve == 2) { // BLS12-381 G1 generator (standard; on-curve and order checked in tests/test_oracle.py)
        static const u32 gx[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                   0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
        static const u32 gy[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                                   0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
        for (int k = 0; k < L && k < 12; k++) {
            cx[k] = gx[k];
            cy[k] = gy[k];
        }
    } else {
        static const u32 gx[12] = {0xb21be9efu, 0xeab9b16eu, 0xffcd394eu, 0xd5481512u, 0xbd37cb5cu, 0x188282c8u,
                                   0xaa9d41bbu, 0x85951e2cu, 0xbf87ff54u, 0xc8fc6225u, 0xfe740a67u, 0x008848deu};
        static const u32 gy[12] = {0x559c8ea6u, 0xfd82de55u, 0x34a9591au, 0xc2fe3d36u, 0x4fb82305u, 0x6d182ad4u,
                                   0xca3e52d9u, 0xbd7fb348u, 0x30afeec4u, 0x1f674f5du, 0xc5102effu, 0x01914a69u};
        for (int k = 0; k < L && k < 12; k++) {
            cx[k] = gx[k];
            cy[k] = gy[k];
        }
    }
    Fe<F> t, k, x;
    fe_const(k, F::K_TOINT);
    fe_unpack(t, cx);
    fe_mul(x, t, k);
    fe_to_wire(out, x);
    fe_unpack(t, cy);
    fe_mul(x, t, k);
    fe_to_wire(out + L, x);
}

// BN254 G2 generator (the standard one of EIP-197 / arkworks; on the twist y^2 = x^3 + 3/(9+u) and of order r: tests/pyref.py), x = x0 + x1 u
template <>
void generator_wire<Ext2<Bn254Fq>>(u32 *out, unsigned)
{
    static const u32 c[4][8] = {{0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu},
                                {0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u},
                                {0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u},
                                {0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u}};
    for (int j = 0; j < 4; j++) {
        Fe<Bn254Fq> t, k, x;
        fe_const(k, Bn254Fq::K_TOINT);
        fe_unpack(t, c[j]);
        fe_mul(x, t, k);
        fe_to_wire(out + 8 * j, x);
    }
}

template <class F>
hipError_t gen_bases(unsigned curve, u64 seed, u64 first, u64 n, void *d_out, hipStream_t stream)
{
    constexpr int N = F::N, L = F::L;
    u32 gw[2 * L];
    generator_wire<F>(gw, curve);
    u32 *d_table = nullptr, *d_gen = nullptr;
    PANDA_TRY(hipMalloc(&d_table, (size_t)8 * 255 * 2 * N * 4));
    PANDA_TRY(hipMalloc(&d_gen, sizeof(gw)));
    PANDA_TRY(hipMemcpyAsync(d_gen, gw, sizeof(gw), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_gen_table<F>, dim3(1), dim3(64), 0, stream, d_table, d_gen);
    hipLaunchKernelGGL(k_gen_bases<F>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, seed, first, n, d_table, (u32 *)d_out);
    PANDA_TRY(hipGetLastError());
    PANDA_TRY(hipStreamSynchronize(stream));
    PANDA_TRY(hipFree(d_table));
    PANDA_TRY(hipFree(d_gen));
    return hipSuccess;
}

} // namespace

extern "C" {

panda_error panda_debug_field_op(unsigned field_id, unsigned op, void *d_r, const void *d_a, const void *d_b, size_t n, panda_stream stream)
{
    if (op > 6 || field_id > 5) return panda_error_invalid_value;
    hipStream_t s = static_cast<hipStream_t>(stream.handle);
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    u32 *r = (u32 *)d_r;
    const u32 *a = (const u32 *)d_a, *b = (const u32 *)d_b;
    switch (field_id) {
    case 0: hipLaunchKernelGGL(k_field_op<Bn254Fq>, grid, block, 0, s, op, r, a, b, n); break;
    case 1: hipLaunchKernelGGL(k_field_op<Bn254Fr>, grid, block, 0, s, op, r, a, b, n); break;
    case 2: hipLaunchKernelGGL(k_field_op<Bls377Fq>, grid, block, 0, s, op, r, a, b, n); break;
    case 3: hipLaunchKernelGGL(k_field_op<Bls377Fr>, grid, block, 0, s, op, r, a, b, n); break;
    case 4: hipLaunchKernelGGL(k_field_op<Bls381Fq>, grid, block, 0, s, op, r, a, b, n); break;
    default: hipLaunchKernelGGL(k_field_op<Bls381Fr>, grid, block, 0, s, op, r, a, b, n); break;
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return static_cast<panda_error>(e);
}

panda_error panda_debug_curve_op(unsigned curve, unsigned op, void *d_r, const void *d_a, const void *d_b, size_t n, panda_stream stream)
{
    if (op > 4 || curve > 3) return panda_error_invalid_value;
    hipStream_t s = static_cast<hipStream_t>(stream.handle);
    dim3 grid((unsigned)((n + 127) / 128)), block(128);
    if (op >= 3) {
        dim3 qgrid((unsigned)((4 * n + 127) / 128));
        if (curve == 3) hipLaunchKernelGGL(k_curve_op_quad<Ext2<Bn254Fq>>, qgrid, block, 0, s, op, (u32 *)d_r, (const u32 *)d_a, (const u32 *)d_b, n);
        else if (curve == 0) hipLaunchKernelGGL(k_curve_op_quad<Bn254Fq>, qgrid, block, 0, s, op, (u32 *)d_r, (const u32 *)d_a, (const u32 *)d_b, n);
        else if (curve == 1) hipLaunchKernelGGL(k_curve_op_quad<Bls377Fq>, qgrid, block, 0, s, op, (u32 *)d_r, (const u32 *)d_a, (const u32 *)d_b, n);
        else hipLaunchKernelGGL(k_curve_op_quad<Bls381Fq>, qgrid, block, 0, s, op, (u32 *)d_r, (const u32 *)d_a, (const u32 *)d_b, n);
    } else if (curve == 3) hipLaunchKernelGGL(k_curve_op<Ext2<Bn254Fq>>, grid, block, 0, s, op, (u32 *)d_r, (const u32 *)d_a, (const u32 *)d_b, n);
    else if (curve == 0) hipLaunchKernelGGL(k_curve_op<Bn254Fq>, grid, block, 0, s, op, (u32 *)d_r, (const u32 *)d_a, (const u32 *)d_b, n);
    else if (curve == 1) hipLaunchKernelGGL(k_curve_op<Bls377Fq>, grid, block, 0, s, op, (u32 *)d_r, (const u32 *)d_a, (const u32 *)d_b, n);
    else hipLaunchKernelGGL(k_curve_op<Bls381Fq>, grid, block, 0, s, op, (u32 *)d_r, (const u32 *)d_a, (const u32 *)d_b, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return static_cast<panda_error>(e);
}

panda_error panda_gen_scalars(unsigned curve, uint64_t seed, uint64_t first, uint64_t n, void *d_out, panda_stream stream)
{
    if (curve > 3) return panda_error_invalid_value;
    hipStream_t s = static_cast<hipStream_t>(stream.handle);
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (curve == 0 || curve == 3) hipLaunchKernelGGL(k_gen_scalars<Bn254Fr>, grid, block, 0, s, seed, first, n, (u32 *)d_out);
    else if (curve == 1) hipLaunchKernelGGL(k_gen_scalars<Bls377Fr>, grid, block, 0, s, seed, first, n, (u32 *)d_out);
    else hipLaunchKernelGGL(k_gen_scalars<Bls381Fr>, grid, block, 0, s, seed, first, n, (u32 *)d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return static_cast<panda_error>(e);
}

panda_error panda_gen_bases(unsigned curve, uint64_t seed, uint64_t first, uint64_t n, void *d_out, panda_stream stream)
{
    if (curve > 3) return panda_error_invalid_value;
    hipStream_t s = static_cast<hipStream_t>(stream.handle);
    switch (curve) {
    case 0: return static_cast<panda_error>(gen_bases<Bn254Fq>(curve, seed, first, n, d_out, s));
    case 1: return static_cast<panda_error>(gen_bases<Bls377Fq>(curve, seed, first, n, d_out, s));
    case 2: return static_cast<panda_error>(gen_bases<Bls381Fq>(curve, seed, first, n, d_out, s));
    default: return static_cast<panda_error>(gen_bases<Ext2<Bn254Fq>>(curve, seed, first, n, d_out, s));
    }
}

} // extern "C"
