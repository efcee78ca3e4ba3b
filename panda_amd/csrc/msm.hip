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
        // ordered against the caller's stream by events only.  PANDA_HELPER_PRIORITY (experiments): a priority for the stream
        const char *prio = getenv("PANDA_HELPER_PRIORITY");
        e = prio ? hipStreamCreateWithPriority(&g_helper.s, hipStreamNonBlocking, atoi(prio)) : hipStreamCreateWithFlags(&g_helper.s, hipStreamNonBlocking);
        if (e != hipSuccess) {
            g_helper.s = nullptr;
            return e;
        }
        g_helper.device = dev;
    }
    *out = g_helper.s;
    return hipSuccess;
}

namespace {
struct Mailbox {
    uint32_t *words = nullptr;
    void drop()
    {
        if (words) (void)hipHostFree(words);
        words = nullptr;
    }
    ~Mailbox() { drop(); }
};
thread_local Mailbox g_mailbox;
} // namespace

hipError_t thread_mailbox(uint32_t **host_words)
{
    if (!g_mailbox.words) {
        // portable + mapped: one address for the host and for every device
        hipError_t e = hipHostMalloc((void **)&g_mailbox.words, MAILBOX_WORDS * sizeof(uint32_t), hipHostMallocPortable | hipHostMallocMapped);
        if (e != hipSuccess) {
            g_mailbox.words = nullptr;
            return e;
        }
    }
    *host_words = g_mailbox.words;
    return hipSuccess;
}

namespace {
std::atomic<unsigned> g_clock_stamps{0};
thread_local ClockDelta g_msm_clock, g_ntt_clock;

__global__ void k_clock_stamp(ulonglong2 *block)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const unsigned long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
        block[(xcc & 7u) * 256u + ((hw >> 8) & 255u)] = make_ulonglong2(t, r); // one 16-byte store: waves of one CU may race, any of them will do
    }
}

struct StampBlocks {
    uint64_t *words = nullptr;
    void drop()
    {
        if (words) (void)hipHostFree(words);
        words = nullptr;
    }
    ~StampBlocks() { drop(); }
};
thread_local StampBlocks g_stamp_blocks;
} // namespace

bool clock_stamps_enabled() { return g_clock_stamps.load(std::memory_order_relaxed) != 0; }
void set_clock_stamps_enabled(bool on) { g_clock_stamps.store(on ? 1u : 0u, std::memory_order_relaxed); }

hipError_t thread_stamp_blocks(uint64_t **host_blocks)
{
    if (!g_stamp_blocks.words) {
        hipError_t e = hipHostMalloc((void **)&g_stamp_blocks.words, 4 * CLOCK_STAMP_SLOTS * 2 * sizeof(uint64_t), hipHostMallocPortable | hipHostMallocMapped);
        if (e != hipSuccess) {
            g_stamp_blocks.words = nullptr;
            return e;
        }
    }
    *host_blocks = g_stamp_blocks.words;
    return hipSuccess;
}

hipError_t enqueue_clock_stamp(hipStream_t s, uint64_t *block)
{
    hipLaunchKernelGGL(k_clock_stamp, dim3(2048), dim3(64), 0, s, (ulonglong2 *)block);
    return hipGetLastError();
}

ClockDelta clock_delta(const uint64_t *before, const uint64_t *after)
{
    ClockDelta d;
    std::vector<uint64_t> all_ticks;
    uint64_t sum = 0;
    for (unsigned x = 0; x < 8; x++) {
        std::vector<uint64_t> cyc;
        for (unsigned k = 0; k < 256; k++) {
            const uint64_t *b = before + (size_t)(x * 256 + k) * 2, *a = after + (size_t)(x * 256 + k) * 2;
            if (b[1] && a[1] > b[1] && a[0] > b[0]) {
                cyc.push_back(a[0] - b[0]);
                all_ticks.push_back(a[1] - b[1]);
            }
        }
        if (cyc.empty()) continue;
        std::sort(cyc.begin(), cyc.end());
        d.per_xcd[x] = cyc[cyc.size() / 2];
        sum += d.per_xcd[x];
        d.xcds++;
        if (!d.cycles || d.per_xcd[x] < d.cycles) d.cycles = d.per_xcd[x];
    }
    if (d.xcds) {
        d.cycles_mean = sum / d.xcds;
        std::sort(all_ticks.begin(), all_ticks.end());
        d.ticks = all_ticks[all_ticks.size() / 2];
    }
    return d;
}

void clock_delta_out(const ClockDelta &d, uint64_t *out)
{
    out[0] = d.cycles;
    out[1] = d.ticks;
    out[2] = d.xcds;
    out[3] = d.cycles_mean;
    for (int x = 0; x < 8; x++) out[4 + x] = d.per_xcd[x];
}

ClockDelta &thread_msm_clock() { return g_msm_clock; }
ClockDelta &thread_ntt_clock() { return g_ntt_clock; }

hipError_t release_thread_arena()
{
    g_helper.drop();
    g_mailbox.drop();
    g_stamp_blocks.drop();
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
// at registration: the first kernel of every execute compares them (msm_sort.hip, check_samples) and, on a mismatch, k_accumulate
// skips its work, the entry is forgotten and the call runs again from the caller's buffer.
// What is and is not guaranteed: the per-call comparison is a SAMPLE (REG_SAMPLES rows) -- it catches a buffer that was freed and
// reused wholesale, not one whose content changed in unsampled rows only.  The STRICT check is the 64-bit hash of the whole wire
// buffer taken at registration: panda_msm_verify_registered() recomputes it on demand (one streaming pass, 0.2 ms per GiB), and
// panda_msm_set_paranoid(1) does so in front of every execute.  Callers that recycle device buffers behind the library's back
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
std::atomic<unsigned> g_phase_timing{0};
std::atomic<unsigned> g_paranoid{0};
constexpr unsigned kOverlapAuto = 0xffffffffu;
std::atomic<unsigned> g_overlap_front{kOverlapAuto};
std::atomic<unsigned> g_overlap_wgs{0};
std::atomic<unsigned> g_acc_variant{0};
std::atomic<unsigned> g_chunk_first{1};

// With tables, the accumulation of the first `front / 128` of the bucket space can run beside the sort of the rest (msm_impl.h,
// "want_split").  Measured in round 5 and NOT the policy (profiles/r05_overlap_sort_accumulate.txt): k_accumulate<Bn254Fq> needs its
// four waves per SIMD (three: -22 %), and those hold 94 % of the register file, so the second stream's sort workgroups are only
// dispatched once the accumulate grid has nothing left to place (+0.6 ms at 2^24); a grid small enough to leave them room costs the
// accumulate more than the 1.7 ms of sort it hides.  The option stays for experiments and for fields whose accumulate leaves room.
unsigned pick_overlap_front(unsigned)
{
    const unsigned forced = g_overlap_front.load(std::memory_order_relaxed);
    return forced != kOverlapAuto ? forced : 0u;
}

const char *const kPhaseNames[PANDA_MSM_PHASES] = {"convert_bases+digits", "sort_partition", "sort_buckets", "accumulate",
                                                   "fixup", "bucket_reduce", "d2h+host_horner", "total_device"};

// A bucket costs about as much as five mixed additions by the time it has been through the fix-up, the row / column sums and the
// finishing kernel (fitted to 2^20 ... 2^24 points with and without tables: profiles/r04_window_sweeps.txt).
constexpr double kBucketCost = 5.0;
// The same quantity on the plain path, fitted separately (ibid.): 4.5.  There every window has a bucket space of its own, a quarter to a
// sixteenth the size of the tables' shared one, so the row / column sums are shorter lines and the fix-up has fewer pieces per bucket.
constexpr double kBucketCostPlain = 4.5;

// window width policy of the plain path (replaces get_window_bits_count, msm_cuda.cuh:21-45): every window has its own 2^(c-1) buckets
// (signed digits), so minimise  W(c) * (n * s(c) + kBucketCostPlain * 2^(c-1))  over the widths the sort supports, s = 1.04 for the windows wider
// than 16 bits (their three-level sort moves every entry once more): 16 bits up to 2^22 points, 17 at 2^23, 20 from 2^24 on
// (measured: 2^24 21.4 -> 19.7 ms, 2^23 11.09 -> 10.8; rounds 1-3 stopped at 16 bits: two-byte digit codes and a two-level sort)
unsigned pick_window_bits(unsigned fr, unsigned log_n)
{
    if (const unsigned forced = g_window_override.load(std::memory_order_relaxed)) {
        const unsigned c = std::min(std::max(forced, 4u), 20u);
        return panda::msm_sort_plain_supported(log_n, panda::make_safe_window_plan(fr, c)) ? c : std::min(c, 16u);
    }
    unsigned best = 0;
    double best_cost = 0;
    for (unsigned c = 4; c <= 20; c++) {
        const panda::WindowPlan plan = panda::make_safe_window_plan(fr, c);
        if (plan.width[0] != c || !panda::msm_sort_plain_supported(log_n, plan)) continue;
        const double cost = (double)plan.W * ((double)((u64)1 << log_n) * (c > 16 ? 1.04 : 1.0) + kBucketCostPlain * (double)(1u << (c - 1)));
        if (!best || cost < best_cost) {
            best = c;
            best_cost = cost;
        }
    }
    return best ? best : 4u;
}

// window width for precomputed tables: all windows share one bucket space, so wide windows are cheap --
// minimise (additions) n * W(c) + kBucketCost * 2^(c-1) over the widths the sort supports (2^20: 19 bits, 2^21 / 2^22: 20, 2^23 / 2^24: 22)
unsigned pick_tabled_window_bits(unsigned fr, unsigned log_n)
{
    unsigned best = 0;
    double best_cost = 0;
    for (unsigned c = 10; c <= 24; c++) {
        const panda::WindowPlan plan = panda::make_safe_window_plan(fr, c);
        if (plan.width[0] != c || !panda::msm_sort_tabled_supported(log_n, plan)) continue;
        const double cost = (double)plan.W * (double)((u64)1 << log_n) + kBucketCost * (double)(1u << (c - 1));
        if (!best || cost < best_cost) {
            best = c;
            best_cost = cost;
        }
    }
    return best;
}

hipError_t msm_execute_on(unsigned curve, const panda_msm_configuration &cfg, const panda::MsmRegistration *r, bool *stale, const panda::MsmPipeline *pipe)
{
    const panda::MsmTuning tuning{pick_window_bits(panda::msm_scalar_field_of(curve), cfg.log_scalars_count), g_chunk.load(std::memory_order_relaxed),
                                  g_phase_timing.load(std::memory_order_relaxed), pick_overlap_front(cfg.log_scalars_count), g_overlap_wgs.load(std::memory_order_relaxed),
                                  g_chunk_first.load(std::memory_order_relaxed), g_acc_variant.load(std::memory_order_relaxed)};
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
    RegisteredPtr reg = lookup_registered(cfg.bases, cfg.log_scalars_count, curve); // held for the whole call
    if (reg && g_paranoid.load(std::memory_order_relaxed)) { // strict mode: the whole buffer must still hash to what was registered
        uint64_t now = 0;
        PANDA_TRY(panda::msm_hash_wire(cfg.bases, n * kAffineBytes[curve], static_cast<hipStream_t>(cfg.stream.handle), &now));
        if (now != reg->hash) {
            fprintf(stderr, "[panda-hip] registered bases at %p changed since registration (hash): registration dropped, converting per call\n", cfg.bases);
            const RegisteredBases *gone = reg.get();
            forget_if([gone](const RegisteredBases &r) { return &r == gone; });
            reg.reset();
        }
    }
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

namespace {

__device__ __forceinline__ u64 mix64(u64 x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// sum over the 16-byte units of the buffer of a position-dependent mix of their content: order-independent, so every workgroup adds
// its share with one atomic; one streaming pass (0.2 ms per GiB)
__global__ void __launch_bounds__(256) k_hash_wire(const uint4 *__restrict__ buf, u64 units, unsigned long long *__restrict__ out)
{
    __shared__ u64 partial[4];
    u64 h = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < units; i += (u64)gridDim.x * 256) {
        const uint4 v = buf[i];
        const u64 a = ((u64)v.y << 32) | v.x, b = ((u64)v.w << 32) | v.z;
        h += mix64(a + 0x9E3779B97F4A7C15ull * (2 * i + 1)) ^ mix64(b ^ (0xD1342543DE82EF95ull * (2 * i + 2)));
    }
    for (int d = 32; d > 0; d >>= 1) h += __shfl_down(h, d, 64);
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(partial[0] + partial[1] + partial[2] + partial[3]));
}

} // namespace

namespace panda {

hipError_t msm_hash_wire(const void *d_buf, size_t bytes, hipStream_t s, uint64_t *hash)
{
    // the workgroups add into a word of DEVICE memory (atomics on host memory would need PCIe atomics, which not every platform
    // routes); it is copied into the pinned mailbox behind the kernel
    uint32_t *mail = nullptr;
    PANDA_TRY(thread_mailbox(&mail));
    unsigned long long *h_sum = reinterpret_cast<unsigned long long *>(mail + 32);
    unsigned long long *d_sum = nullptr;
    PANDA_TRY(hipMalloc((void **)&d_sum, 8));
    hipError_t e = hipMemsetAsync(d_sum, 0, 8, s);
    const u64 units = bytes / 16;
    const unsigned blocks = (unsigned)std::min<u64>((units + 255) / 256, 4096);
    if (e == hipSuccess && units) {
        hipLaunchKernelGGL(k_hash_wire, dim3(blocks), dim3(256), 0, s, (const uint4 *)d_buf, units, d_sum);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_sum, d_sum, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_sum);
    if (e != hipSuccess) return e;
    *hash = *h_sum;
    return hipSuccess;
}

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

panda_error panda_msm_verify_registered(const void *d_bases, panda_stream stream)
{
    RegisteredPtr reg;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        for (const auto &r : g_registry)
            if (r->wire == d_bases) reg = r;
    }
    if (!reg) return panda_error_invalid_value;
    uint64_t now = 0;
    const hipError_t e = panda::msm_hash_wire(d_bases, ((size_t)1 << reg->log_n) * kAffineBytes[reg->curve], static_cast<hipStream_t>(stream.handle), &now);
    if (e != hipSuccess) return static_cast<panda_error>(e);
    if (now == reg->hash) return panda_success;
    const RegisteredBases *gone = reg.get();
    forget_if([gone](const RegisteredBases &r) { return &r == gone; });
    return panda_error_invalid_value; // the buffer no longer holds what was registered: the registration is gone
}

panda_error panda_msm_set_paranoid(unsigned on)
{
    g_paranoid.store(on ? 1u : 0u, std::memory_order_relaxed);
    return panda_success;
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
    if (window_bits > 20) return panda_error_invalid_value;
    g_window_override.store(window_bits, std::memory_order_relaxed);
    return panda_success;
}

panda_error panda_msm_plain_window_plan(unsigned curve, unsigned log_n, unsigned *window_bits, unsigned *windows)
{
    if (curve > 3 || log_n > 26) return panda_error_invalid_value;
    const unsigned fr = panda::msm_scalar_field_of(curve);
    const panda::WindowPlan plan = panda::make_safe_window_plan(fr, pick_window_bits(fr, log_n));
    if (window_bits) *window_bits = plan.width[0];
    if (windows) *windows = plan.W;
    return panda_success;
}

panda_error panda_msm_set_overlap(unsigned front_of_128, unsigned workgroups_per_cu)
{
    if ((front_of_128 >= 128 && front_of_128 != kOverlapAuto) || workgroups_per_cu > 64) return panda_error_invalid_value;
    g_overlap_front.store(front_of_128, std::memory_order_relaxed);
    g_overlap_wgs.store(workgroups_per_cu, std::memory_order_relaxed);
    return panda_success;
}

panda_error panda_msm_set_accumulate_variant(unsigned variant)
{
    if (variant > 5) return panda_error_invalid_value;
    g_acc_variant.store(variant, std::memory_order_relaxed);
    return panda_success;
}

panda_error panda_msm_set_chunk_first(unsigned on)
{
    g_chunk_first.store(on ? 1u : 0u, std::memory_order_relaxed);
    return panda_success;
}

panda_error panda_msm_set_wide_merge(unsigned mode)
{
    if (mode > 3) return panda_error_invalid_value;
    panda::msm_sort_set_wide_merge(mode);
    return panda_success;
}

panda_error panda_msm_set_reduce_group(unsigned) { return panda_success; } // round-3 knob of a kernel that no longer exists

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

panda_error panda_set_clock_stamps(unsigned on)
{
    panda::set_clock_stamps_enabled(on != 0);
    return panda_success;
}

panda_error panda_msm_last_clock(uint64_t *out)
{
    if (!out) return panda_error_invalid_value;
    panda::clock_delta_out(panda::thread_msm_clock(), out);
    return panda_success;
}

panda_error panda_clock_stamp(panda_stream stream, void *block)
{
    if (!block) return panda_error_invalid_value;
    return static_cast<panda_error>(panda::enqueue_clock_stamp(static_cast<hipStream_t>(stream.handle), (uint64_t *)block));
}

panda_error panda_clock_delta(const void *before, const void *after, uint64_t *out)
{
    if (!before || !after || !out) return panda_error_invalid_value;
    panda::clock_delta_out(panda::clock_delta((const uint64_t *)before, (const uint64_t *)after), out);
    return panda_success;
}

const char *panda_msm_phase_name(unsigned phase) { return phase < PANDA_MSM_PHASES ? kPhaseNames[phase] : ""; }

} // extern "C"
