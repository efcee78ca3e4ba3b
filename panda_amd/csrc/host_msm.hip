// host_msm.hip -- the CPU halves of the C ABI: panda_msm_execute_*_host and panda_msm_combine_*.
//
// panda_msm_execute_bn254_host replaces the reference's single-threaded host-debug Pippenger
// (src/cuda/core/unit/msm/msm_host.cuh:267-383, entered from panda_interface.cu:162-165; every pointer in
// the configuration is a HOST pointer, unit.rs:374-393).  It runs the same fe29/curve29 arithmetic as the
// device kernels, compiled for the host, with signed-digit buckets; unlike the reference it leaves the
// caller's scalar buffer untouched (msm_host.cuh:293-296 converts it in place).
// This is product code and does not use anything under oracle/.
#include <algorithm>
#include <string.h>
#include <vector>

#include "curve29.h"
#include "panda_internal.h"

using namespace panda29;

namespace {

template <class Fq, class Fr>
int host_msm(const panda_msm_configuration &cfg)
{
    constexpr int LQ = Fq::L, LR = Fr::L;
    if (!cfg.bases || !cfg.scalars || !cfg.results || cfg.log_scalars_count > 30) return hipErrorInvalidValue;
    const unsigned log_n = cfg.log_scalars_count;
    const size_t n = (size_t)1 << log_n;
    const unsigned c = (unsigned)std::min(std::max((int)log_n - 3, 3), 13);
    const unsigned W = (Fr::BITS + 1 + c - 1) / c;
    const unsigned NB = 1u << (c - 1);
    const u32 *bases = (const u32 *)cfg.bases;
    const u32 *scalars = (const u32 *)cfg.scalars;

    // canonical scalars and internal-form bases, once
    std::vector<u32> canon(n * (LR + 1));
    std::vector<Fe<Fq>> bx(n), by(n);
    std::vector<unsigned char> binf(n);
    for (size_t i = 0; i < n; i++) {
        fe_wire_to_canonical<Fr>(&canon[i * (LR + 1)], scalars + i * LR);
        canon[i * (LR + 1) + LR] = 0;
        binf[i] = affine_from_wire(bx[i], by[i], bases + i * 2 * LQ);
    }
    std::vector<unsigned char> carry(n, 0);
    std::vector<Xyzz<Fq>> buckets(NB), windows(W);
    for (unsigned w = 0; w < W; w++) {
        for (auto &b : buckets) xyzz_set_identity(b);
        const unsigned lo = w * c, m = lo >> 5, sh = lo & 31;
        for (size_t i = 0; i < n; i++) {
            const u32 *s = &canon[i * (LR + 1)];
            u32 raw = 0;
            if (m < (unsigned)LR) raw = (u32)((((u64)s[m + 1] << 32) | s[m]) >> sh) & ((1u << c) - 1);
            raw += carry[i];
            if (raw >= NB) { // negative digit raw - 2^c
                carry[i] = 1;
                u32 mag = (1u << c) - raw;
                if (mag) {
                    Fe<Fq> ny;
                    fe_neg<Fq, 2>(ny, by[i]);
                    xyzz_madd(buckets[mag - 1], bx[i], ny, binf[i]);
                }
            } else {
                carry[i] = 0;
                if (raw) xyzz_madd(buckets[raw - 1], bx[i], by[i], binf[i]);
            }
        }
        Xyzz<Fq> run, sum;
        xyzz_set_identity(run);
        xyzz_set_identity(sum);
        for (int b = (int)NB - 1; b >= 0; b--) {
            xyzz_add(run, buckets[b]);
            xyzz_add(sum, run);
        }
        windows[w] = sum;
    }
    Xyzz<Fq> acc, d;
    xyzz_set_identity(acc);
    for (int w = (int)W - 1; w >= 0; w--) {
        for (unsigned k = 0; k < c; k++) {
            xyzz_dbl(d, acc);
            acc = d;
        }
        xyzz_add(acc, windows[w]);
    }
    u32 out[3 * LQ];
    if (cfg.msm_result_coordinate_type == PROJECTIVE)
        xyzz_to_homogeneous_wire(out, acc);
    else
        xyzz_to_jacobian_wire(out, acc);
    memcpy(cfg.results, out, sizeof(out));
    return hipSuccess;
}

// sum of `count` Jacobian wire partials (host or device memory) -> one point on the host
template <class Fq>
int combine(const void *partials, unsigned count, panda_msm_result_coordinate_type out_type, void *result)
{
    constexpr int LQ = Fq::L;
    if (!partials || !result) return hipErrorInvalidValue;
    std::vector<u32> h((size_t)count * 3 * LQ);
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices == 0) {
        (void)hipGetLastError();
        memcpy(h.data(), partials, h.size() * 4); // no device in this process: the partials can only be host memory
    } else {
        hipError_t e = hipMemcpy(h.data(), partials, h.size() * 4, hipMemcpyDefault);
        if (e != hipSuccess) return e;
    }
    Xyzz<Fq> acc, q;
    xyzz_set_identity(acc);
    for (unsigned i = 0; i < count; i++) {
        xyzz_from_jacobian_wire(q, h.data() + (size_t)i * 3 * LQ);
        xyzz_add(acc, q);
    }
    u32 out[3 * LQ];
    if (out_type == PROJECTIVE)
        xyzz_to_homogeneous_wire(out, acc);
    else
        xyzz_to_jacobian_wire(out, acc);
    memcpy(result, out, sizeof(out));
    return hipSuccess;
}

} // namespace

extern "C" {

panda_error panda_msm_execute_bn254_host(const panda_msm_configuration cfg) { return static_cast<panda_error>(host_msm<Bn254Fq, Bn254Fr>(cfg)); }

panda_error panda_msm_execute_bls12_377_host(const panda_msm_configuration cfg) { return static_cast<panda_error>(host_msm<Bls377Fq, Bls377Fr>(cfg)); }

panda_error panda_msm_execute_bls12_381_host(const panda_msm_configuration cfg) { return static_cast<panda_error>(host_msm<Bls381Fq, Bls381Fr>(cfg)); }

panda_error panda_msm_execute_bn254_g2_host(const panda_msm_configuration cfg) { return static_cast<panda_error>(host_msm<Ext2<Bn254Fq>, Bn254Fr>(cfg)); }

panda_error panda_msm_combine_bn254_g2(const void *partials, unsigned count, panda_msm_result_coordinate_type out_type, void *result)
{
    return static_cast<panda_error>(combine<Ext2<Bn254Fq>>(partials, count, out_type, result));
}

panda_error panda_msm_combine_bn254(const void *partials, unsigned count, panda_msm_result_coordinate_type out_type, void *result)
{
    return static_cast<panda_error>(combine<Bn254Fq>(partials, count, out_type, result));
}

panda_error panda_msm_combine_bls12_377(const void *partials, unsigned count, panda_msm_result_coordinate_type out_type, void *result)
{
    return static_cast<panda_error>(combine<Bls377Fq>(partials, count, out_type, result));
}

panda_error panda_msm_combine_bls12_381(const void *partials, unsigned count, panda_msm_result_coordinate_type out_type, void *result)
{
    return static_cast<panda_error>(combine<Bls381Fq>(partials, count, out_type, result));
}

} // extern "C"
