// msm.hip -- curve-independent host side of the MSM: scratch arena, cached-bases registry, window policies, C ABI.
// The kernels and the per-call driver live in msm_impl.h (instantiated per curve in msm_bn254.hip / msm_bls377.hip),
// the digit extraction and the bucket sort in msm_sort.hip.
#include <algorithm>
#include <atomic>
#include <memory>
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

Arena::~Arena()
{
    // a host thread that ends without calling panda_msm_tear_down() must not leak its scratch (several GiB at 2^24)
    if (base) (void)hipFree(base);
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

namespace {
struct HelperStream {
    hipStream_t s = nullptr;
    int device = -1;
    void drop()
    {
        if (s) (void)hipStreamDestroy(s);
        s = nullptr;
        device = -1;
    }
    ~HelperStream() { drop(); }
};
thread_local HelperStream g_helper;
} // namespace

hipError_t thread_helper_stream(hipStream_t *out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (g_helper.s && g_helper.device != dev) g_helper.drop();
    if (!g_helper.s) {
        e = hipStreamCreateWithFlags(&g_helper.s, hipStreamNonBlocking); // ordered against the caller's stream by events only
        if (e != hipSuccess) {
            g_helper.s = nullptr;
            return e;
        }
        g_helper.device = dev;
    }
    *out = g_helper.s;
    return hipSuccess;
}

hipError_t release_thread_arena()
{
    g_helper.drop();
    return thread_arena().release();
}

} // namespace panda

namespace {

thread_local float g_phase_ms[PANDA_MSM_PHASES] = {0};

// Cached bases (README "Supports cached bases and scalars"; init_msm, wrapper.rs:122-152): a caller that keeps a base
// set on the device for many MSMs may register it; the radix conversion k_convert_bases would repeat on every call is
// then done once and kept next to it -- optionally together with the window tables above.  The caller promises not to
// modify a registered buffer until it is unregistered.
//
// Lifetime: an entry owns its device copy and is handed out as a shared_ptr, so a call that is executing keeps the
// tables alive even if another host thread unregisters them meanwhile; the memory goes when the last user lets go.
// Staleness: the key is the caller's raw device address, which an allocator may hand out again.  panda_free /
// panda_free_async drop every entry whose buffer lies in the allocation being freed, and for buffers freed behind the
// library's back (a caching allocator such as torch's) each entry keeps REG_SAMPLES rows of the wire buffer as it was
// at registration: every execute compares them on the device and, on a mismatch, forgets the entry and runs the call
// again from the caller's buffer.  The comparison is a SAMPLE (REG_SAMPLES rows): a recycled address whose new content differs
// from the old one in unsampled rows only is not detected -- callers that recycle device buffers behind the library's back
// should unregister first; panda_free / panda_free_async do it for them.
struct RegisteredBases : panda::MsmRegistration {
    RegisteredBases() : panda::MsmRegistration{} {}
    RegisteredBases(const RegisteredBases &) = delete;
    RegisteredBases &operator=(const RegisteredBases &) = delete;
    ~RegisteredBases()
    {
        if (converted) (void)hipFree(converted); // waits for the device: nothing can still be reading the tables
    }
};
typedef std::shared_ptr<RegisteredBases> RegisteredPtr;
std::mutex g_registry_mutex;
// deliberately never destroyed: entries still registered when the process exits would otherwise call hipFree from a static
// destructor, after the HIP runtime's own exit handler may already have run (the OS reclaims the memory)
std::vector<RegisteredPtr> &g_registry = *new std::vector<RegisteredPtr>();
std::atomic<size_t> g_registry_count{0}; // lets panda_free skip the lock when nothing is registered

RegisteredPtr lookup_registered(const void *wire, unsigned log_n, unsigned curve)
{
    int dev = -1;
    if (g_registry_count.load(std::memory_order_acquire) == 0 || hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    for (const auto &r : g_registry)
        if (r->wire == wire && r->log_n == log_n && r->curve == curve && r->device == dev) return r;
    return nullptr;
}

// removes the entries `pred` selects; their device memory is released after the lock is dropped, and only once no
// executing call holds them any more
template <class Pred>
size_t forget_if(Pred pred)
{
    std::vector<RegisteredPtr> dropped;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        for (size_t i = 0; i < g_registry.size();) {
            if (pred(*g_registry[i])) {
                dropped.push_back(std::move(g_registry[i]));
                g_registry.erase(g_registry.begin() + i);
            } else
                i++;
        }
        g_registry_count.store(g_registry.size(), std::memory_order_release);
    }
    return dropped.size();
}

std::atomic<unsigned> g_window_override{0};
std::atomic<unsigned> g_chunk{0};
std::atomic<unsigned> g_phase_timing{1};

const char *const kPhaseNames[PANDA_MSM_PHASES] = {"convert_bases+digits", "sort_partition", "sort_buckets", "accumulate",
                                                   "fixup", "bucket_reduce", "d2h+host_horner", "total_device"};

// window width policy (replaces get_window_bits_count, msm_cuda.cuh:21-45)
unsigned pick_window_bits(unsigned log_n)
{
    const unsigned forced = g_window_override.load(std::memory_order_relaxed);
    if (forced) return std::min(std::max(forced, 4u), 16u);
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

hipError_t msm_execute_on(unsigned curve, const panda_msm_configuration &cfg, const panda::MsmRegistration *r, bool *stale, const panda::MsmPipeline *pipe)
{
    const panda::MsmTuning tuning{pick_window_bits(cfg.log_scalars_count), g_chunk.load(std::memory_order_relaxed),
                                  g_phase_timing.load(std::memory_order_relaxed)};
    switch (curve) {
    case 0: return panda::msm_execute_bn254(cfg, r, tuning, g_phase_ms, stale, pipe);
    case 1: return panda::msm_execute_bls377(cfg, r, tuning, g_phase_ms, stale, pipe);
    case 2: return panda::msm_execute_bls381(cfg, r, tuning, g_phase_ms, stale, pipe);
    default: return panda::msm_execute_bn254_g2(cfg, r, tuning, g_phase_ms, stale, pipe);
    }
}

// bytes of one affine base / one result on the wire (2 / 3 coordinates of L 32-bit limbs; G2 coordinates are pairs)
constexpr size_t kAffineBytes[4] = {64, 96, 96, 128};
constexpr size_t kResultBytes[4] = {96, 144, 144, 192};

hipError_t msm_execute(unsigned curve, const panda_msm_configuration &cfg, const panda::MsmPipeline *pipe = nullptr)
{
    if (cfg.log_scalars_count > 26 || !cfg.bases) return hipErrorInvalidValue;
    // buffers shorter than log_scalars_count implies are refused here instead of faulting in a kernel
    const size_t n = (size_t)1 << cfg.log_scalars_count;
    if (panda::extent_too_short(cfg.bases, n * kAffineBytes[curve]) || panda::extent_too_short(cfg.scalars, n * 32) ||
        panda::extent_too_short(cfg.results, kResultBytes[curve]))
        return hipErrorInvalidValue;
    const RegisteredPtr reg = lookup_registered(cfg.bases, cfg.log_scalars_count, curve); // held for the whole call
    bool stale = false;
    hipError_t e = msm_execute_on(curve, cfg, reg.get(), &stale, pipe);
    if (e == hipSuccess && stale) {
        // the buffer no longer holds the bases it held when it was registered (freed and reallocated behind our back):
        // the entry is dropped and the call answered from the caller's buffer as it is now
        fprintf(stderr, "[panda-hip] registered bases at %p changed since registration: registration dropped, converting per call\n", cfg.bases);
        const RegisteredBases *gone = reg.get();
        forget_if([gone](const RegisteredBases &r) { return &r == gone; });
        e = msm_execute_on(curve, cfg, nullptr, nullptr, pipe);
    }
    return e;
}

hipError_t register_bases(unsigned curve, const void *d_bases, unsigned log_n, bool tabled, unsigned window_bits, hipStream_t s)
{
    if (curve > 3 || !d_bases || log_n > 26) return hipErrorInvalidValue;
    if (panda::extent_too_short(d_bases, ((size_t)1 << log_n) * kAffineBytes[curve])) return hipErrorInvalidValue;
    const unsigned fr = panda::msm_scalar_field_of(curve);
    if (const RegisteredPtr have = lookup_registered(d_bases, log_n, curve)) {
        if (!tabled && !have->tabled) return hipSuccess;
        if (tabled && (have->tabled || log_n < 4)) return hipSuccess; // tables exist (or were not worth building)
        return hipErrorInvalidValue; // registered differently: unregister first
    }
    RegisteredPtr r = std::make_shared<RegisteredBases>();
    r->wire = d_bases;
    r->log_n = log_n;
    r->curve = curve;
    r->tabled = tabled;
    PANDA_TRY(hipGetDevice(&r->device));
    if (tabled) {
        if (window_bits && (window_bits < 4 || window_bits > 24)) return hipErrorInvalidValue;
        const unsigned c = window_bits ? window_bits : pick_tabled_window_bits(fr, log_n);
        if (c) r->plan = panda::make_safe_window_plan(fr, c);
        // sizes the three-level sort has no geometry for (a handful of points) keep the converted copy only
        if (!c || !panda::msm_sort_tabled_supported(log_n, r->plan)) r->tabled = false;
    }
    PANDA_TRY(curve == 0   ? panda::msm_build_registration_bn254(*r, s)
              : curve == 1 ? panda::msm_build_registration_bls377(*r, s)
              : curve == 2 ? panda::msm_build_registration_bls381(*r, s)
                           : panda::msm_build_registration_bn254_g2(*r, s));
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    // another host thread may have registered the same buffer while this one was building its tables: keep the first, drop ours
    for (const auto &have : g_registry)
        if (have->wire == d_bases && have->log_n == log_n && have->curve == curve && have->device == r->device) return hipSuccess;
    g_registry.push_back(std::move(r));
    g_registry_count.store(g_registry.size(), std::memory_order_release);
    return hipSuccess;
}

} // namespace

namespace panda {

// panda_free / panda_free_async: no registration may outlive the buffer it was made for
void registry_forget_allocation(const void *ptr)
{
    if (!ptr || g_registry_count.load(std::memory_order_acquire) == 0) return;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, const_cast<void *>(ptr)) != hipSuccess || !base || !size) {
        (void)hipGetLastError(); // not a range the runtime can describe (stream-ordered pool memory): match the address itself
        base = const_cast<void *>(ptr);
        size = 1;
    }
    const char *lo = (const char *)base, *hi = lo + size;
    forget_if([lo, hi](const RegisteredBases &r) { return (const char *)r.wire >= lo && (const char *)r.wire < hi; });
}

} // namespace panda

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
        if (r->wire == d_bases) {
            if (tables) *tables = r->tabled ? r->plan.W : 1u;
            if (window_bits) *window_bits = r->tabled ? r->plan.width[0] : 0u;
            if (bytes) *bytes = r->bytes;
            return panda_success;
        }
    return panda_error_invalid_value;
}

panda_error panda_msm_unregister_bases(const void *d_bases)
{
    // an MSM that is executing with this registration on another host thread keeps it alive until it returns
    return forget_if([d_bases](const RegisteredBases &r) { return r.wire == d_bases; }) ? panda_success : panda_error_invalid_value;
}

panda_error panda_msm_execute_bn254(const panda_msm_configuration cfg) { return static_cast<panda_error>(msm_execute(0, cfg)); }

panda_error panda_msm_execute_bls12_377(const panda_msm_configuration cfg) { return static_cast<panda_error>(msm_execute(1, cfg)); }

panda_error panda_msm_execute_bls12_381(const panda_msm_configuration cfg) { return static_cast<panda_error>(msm_execute(2, cfg)); }

panda_error panda_msm_setup_bn254_g2(void) { return panda_success; }
panda_error panda_msm_execute_bn254_g2(const panda_msm_configuration cfg) { return static_cast<panda_error>(msm_execute(3, cfg)); }

panda_error panda_msm_execute_from_host(unsigned curve, const panda_msm_configuration cfg, const void *h_scalars, unsigned ranges, panda_stream h2d_stream)
{
    if (curve > 3) return panda_error_invalid_value;
    const panda::MsmPipeline pipe{h_scalars, ranges, static_cast<hipStream_t>(h2d_stream.handle)};
    return static_cast<panda_error>(msm_execute(curve, cfg, &pipe));
}

panda_error panda_msm_set_window_bits(unsigned window_bits)
{
    if (window_bits > 16) return panda_error_invalid_value;
    g_window_override.store(window_bits, std::memory_order_relaxed);
    return panda_success;
}

panda_error panda_msm_set_chunk_entries(unsigned entries)
{
    if (entries > 1024) return panda_error_invalid_value;
    g_chunk.store(entries, std::memory_order_relaxed);
    return panda_success;
}

panda_error panda_msm_set_phase_timing(unsigned level)
{
    if (level > 2) return panda_error_invalid_value;
    g_phase_timing.store(level, std::memory_order_relaxed);
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
