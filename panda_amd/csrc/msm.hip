// msm.hip -- curve-independent host side of the MSM: scratch arena, cached-bases registry, window policies, C ABI.
// The kernels and the per-call driver live in msm_impl.h (instantiated per curve in msm_bn254.hip / msm_bls377.hip),
// the digit extraction and the bucket sort in msm_sort.hip.
#include <algorithm>
#include <mutex>
#include <vector>

#include "msm_impl.h"

typedef uint64_t u64;

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

thread_local float g_phase_ms[PANDA_MSM_PHASES] = {0};

// Cached bases (README "Supports cached bases and scalars"; init_msm, wrapper.rs:122-152): a caller that keeps a base
// set on the device for many MSMs may register it; the radix conversion k_convert_bases would repeat on every call is
// then done once and kept next to it -- optionally together with the window tables above.  The caller promises not to
// modify a registered buffer until it is unregistered.
typedef panda::MsmRegistration RegisteredBases;
std::mutex g_registry_mutex;
std::vector<RegisteredBases> g_registry;

bool lookup_registered(RegisteredBases &out, const void *wire, unsigned log_n, unsigned curve)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    for (const auto &r : g_registry)
        if (r.wire == wire && r.log_n == log_n && r.curve == curve && r.device == dev) {
            out = r;
            return true;
        }
    return false;
}
unsigned g_window_override = 0;
unsigned g_reduce_group = 0;

const char *const kPhaseNames[PANDA_MSM_PHASES] = {"convert_bases+digits", "sort_partition", "sort_buckets", "accumulate",
                                                   "fixup", "bucket_reduce", "d2h+host_horner", "total_device"};

// window width policy (replaces get_window_bits_count, msm_cuda.cuh:21-45)
unsigned pick_window_bits(unsigned log_n)
{
    if (g_window_override) return std::min(std::max(g_window_override, 4u), 16u);
    int c = (int)log_n - 4;
    return (unsigned)std::min(std::max(c, 4), 16);
}

// window width for precomputed tables: all windows share one bucket space, so wide windows are cheap --
// minimise (additions) n * W(c) + (bucket reduction) alpha * 2^(c-1) over the widths the sort supports.  A bucket costs
// about 4 additions' worth of time when the reduction kernels are busy (2^23 points and up) and up to 13 when they are
// latency-bound (measured: 2^20 is fastest at 17 bits, 2^22 at 20, 2^23 and 2^24 at 22).
unsigned pick_tabled_window_bits(unsigned fr, unsigned log_n)
{
    const double alpha = log_n >= 23 ? 4.0 : (log_n >= 22 ? 8.0 : 13.0);
    unsigned best = 0;
    double best_cost = 0;
    for (unsigned c = 10; c <= 23; c++) {
        const panda::WindowPlan plan = panda::make_safe_window_plan(fr, c);
        if (plan.width[0] != c || !panda::msm_sort_tabled_supported(log_n, plan)) continue;
        const double cost = (double)plan.W * (double)((u64)1 << log_n) + alpha * (double)(1u << (c - 1));
        if (!best || cost < best_cost) {
            best = c;
            best_cost = cost;
        }
    }
    return best;
}

hipError_t msm_execute(unsigned curve, const panda_msm_configuration &cfg)
{
    if (cfg.log_scalars_count > 26 || !cfg.bases) return hipErrorInvalidValue;
    RegisteredBases reg{};
    const bool registered = lookup_registered(reg, cfg.bases, cfg.log_scalars_count, curve);
    const panda::MsmTuning tuning{pick_window_bits(cfg.log_scalars_count), g_reduce_group};
    const RegisteredBases *r = registered ? &reg : nullptr;
    switch (curve) {
    case 0: return panda::msm_execute_bn254(cfg, r, tuning, g_phase_ms);
    case 1: return panda::msm_execute_bls377(cfg, r, tuning, g_phase_ms);
    default: return panda::msm_execute_bls381(cfg, r, tuning, g_phase_ms);
    }
}

hipError_t register_bases(unsigned curve, const void *d_bases, unsigned log_n, bool tabled, unsigned window_bits, hipStream_t s)
{
    if (curve > 2 || !d_bases || log_n > 26) return hipErrorInvalidValue;
    RegisteredBases r{};
    if (lookup_registered(r, d_bases, log_n, curve)) {
        if (!tabled && !r.tabled) return hipSuccess;
        if (tabled && (r.tabled || log_n < 4)) return hipSuccess; // tables exist (or were not worth building)
        return hipErrorInvalidValue; // registered differently: unregister first
    }
    r = RegisteredBases{};
    r.wire = d_bases;
    r.log_n = log_n;
    r.curve = curve;
    r.tabled = tabled;
    PANDA_TRY(hipGetDevice(&r.device));
    if (tabled) {
        if (window_bits && (window_bits < 4 || window_bits > 24)) return hipErrorInvalidValue;
        const unsigned c = window_bits ? window_bits : pick_tabled_window_bits(curve, log_n);
        if (c) r.plan = panda::make_safe_window_plan(curve, c);
        // sizes the three-level sort has no geometry for (a handful of points) keep the converted copy only
        if (!c || !panda::msm_sort_tabled_supported(log_n, r.plan)) r.tabled = false;
    }
    PANDA_TRY(curve == 0 ? panda::msm_build_registration_bn254(r, s) : (curve == 1 ? panda::msm_build_registration_bls377(r, s) : panda::msm_build_registration_bls381(r, s)));
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    g_registry.push_back(r);
    return hipSuccess;
}

} // namespace

// ------------------------------------------------------------------------------- C ABI

extern "C" {

panda_error panda_msm_setup_bn254(void) { return panda_success; }
panda_error panda_msm_setup_bls12_377(void) { return panda_success; }
panda_error panda_msm_setup_bls12_381(void) { return panda_success; }

panda_error panda_msm_tear_down(void) { return static_cast<panda_error>(panda::release_thread_arena()); }

panda_error panda_msm_register_bases(unsigned curve, const void *d_bases, unsigned log_n, panda_stream stream)
{
    return static_cast<panda_error>(register_bases(curve, d_bases, log_n, false, 0, static_cast<hipStream_t>(stream.handle)));
}

panda_error panda_msm_precompute_bases(unsigned curve, const void *d_bases, unsigned log_n, unsigned window_bits, panda_stream stream)
{
    return static_cast<panda_error>(register_bases(curve, d_bases, log_n, true, window_bits, static_cast<hipStream_t>(stream.handle)));
}

panda_error panda_msm_registered_info(const void *d_bases, unsigned *tables, unsigned *window_bits, size_t *bytes)
{
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    for (const auto &r : g_registry)
        if (r.wire == d_bases) {
            if (tables) *tables = r.tabled ? r.plan.W : 1u;
            if (window_bits) *window_bits = r.tabled ? r.plan.width[0] : 0u;
            if (bytes) *bytes = r.bytes;
            return panda_success;
        }
    return panda_error_invalid_value;
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

panda_error panda_msm_execute_bn254(const panda_msm_configuration cfg) { return static_cast<panda_error>(msm_execute(0, cfg)); }

panda_error panda_msm_execute_bls12_377(const panda_msm_configuration cfg) { return static_cast<panda_error>(msm_execute(1, cfg)); }

panda_error panda_msm_execute_bls12_381(const panda_msm_configuration cfg) { return static_cast<panda_error>(msm_execute(2, cfg)); }

panda_error panda_msm_set_window_bits(unsigned window_bits)
{
    if (window_bits > 16) return panda_error_invalid_value;
    g_window_override = window_bits;
    return panda_success;
}

panda_error panda_msm_set_reduce_group(unsigned group)
{
    if (group > 64) return panda_error_invalid_value;
    g_reduce_group = group;
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
