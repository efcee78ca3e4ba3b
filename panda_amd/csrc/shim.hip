// shim.hip -- the runtime half of the C ABI: device, stream, event and memory wrappers.
//
// One HIP call per function, error code passed through as `panda_error`, exactly the contract of the
// reference's 28 thin wrappers (src/cuda/core/panda_interface.cu:11-154) and of the two helpers in
// src/cuda/core/common/common.cu:11-29 (blocking-sync stream, device memory pool with an unlimited
// release threshold).  Written against the HIP runtime directly; there is no CUDA path.
#include <map>
#include <mutex>

#include "panda_internal.h"

namespace panda {

namespace {
std::mutex g_alloc_mutex;
// deliberately never destroyed (allocations may be freed from static destructors of the host program)
struct Extent {
    size_t bytes;
    const void *pool; // the stream-ordered pool the memory came from, nullptr for panda_malloc
};
std::map<uintptr_t, Extent> &g_allocs = *new std::map<uintptr_t, Extent>();
} // namespace

void track_allocation(const void *ptr, size_t bytes, const void *pool)
{
    if (!ptr) return;
    std::lock_guard<std::mutex> lock(g_alloc_mutex);
    g_allocs[(uintptr_t)ptr] = Extent{bytes, pool};
}

// a destroyed pool takes its memory with it: extents that were never handed back through panda_free_async must not outlive it (another
// allocator may reuse the range for a LARGER buffer, which a stale, smaller extent would refuse)
void untrack_pool(const void *pool)
{
    if (!pool) return;
    std::lock_guard<std::mutex> lock(g_alloc_mutex);
    for (auto it = g_allocs.begin(); it != g_allocs.end();) it = it->second.pool == pool ? g_allocs.erase(it) : std::next(it);
}

void untrack_allocation(const void *ptr)
{
    if (!ptr) return;
    std::lock_guard<std::mutex> lock(g_alloc_mutex);
    g_allocs.erase((uintptr_t)ptr);
}

bool extent_too_short(const void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return false;
    const uintptr_t p = (uintptr_t)ptr;
    std::lock_guard<std::mutex> lock(g_alloc_mutex);
    auto it = g_allocs.upper_bound(p); // first allocation that starts behind p
    if (it == g_allocs.begin()) return false;
    --it;
    const uintptr_t base = it->first;
    const size_t size = it->second.bytes;
    if (p - base >= size) return false; // not inside an allocation of ours
    return size - (p - base) < bytes;
}

} // namespace panda

extern "C" {

const char *panda_version(void) { return "panda-hip 0.1 (gfx950)"; }

panda_error panda_get_device_number(int *count) { return static_cast<panda_error>(hipGetDeviceCount(count)); }

panda_error panda_get_device(int *device_id) { return static_cast<panda_error>(hipGetDevice(device_id)); }

panda_error panda_set_device(int device_id) { return static_cast<panda_error>(hipSetDevice(device_id)); }

panda_error panda_stream_create(panda_stream *stream, bool blocking_sync)
{
    // HIP has no per-stream synchronisation-policy attribute worth setting here: hipStreamSynchronize
    // on ROCm already blocks the calling thread on an interrupt-driven signal, which is what the
    // reference asks of CUDA with cudaSyncPolicyBlockingSync (common.cu:11-21).
    // Flags stay hipStreamDefault ("blocking" with respect to the NULL stream) because the reference creates
    // its streams with plain cudaStreamCreate and its callers rely on that ordering (unit.rs:418-479).
    (void)blocking_sync;
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamDefault);
    stream->handle = s;
    return static_cast<panda_error>(e);
}

panda_error panda_stream_wait_event(panda_stream stream, panda_event event)
{
    return static_cast<panda_error>(hipStreamWaitEvent(static_cast<hipStream_t>(stream.handle), static_cast<hipEvent_t>(event.handle), 0));
}

panda_error panda_stream_sync(panda_stream stream) { return static_cast<panda_error>(hipStreamSynchronize(static_cast<hipStream_t>(stream.handle))); }

panda_error panda_stream_synchronize(panda_stream stream) { return panda_stream_sync(stream); }

panda_error panda_stream_query(panda_stream stream) { return static_cast<panda_error>(hipStreamQuery(static_cast<hipStream_t>(stream.handle))); }

panda_error panda_stream_destroy(panda_stream stream) { return static_cast<panda_error>(hipStreamDestroy(static_cast<hipStream_t>(stream.handle))); }

panda_error panda_launch_host_fn(panda_stream stream, panda_host_fn fn, void *user_data)
{
    return static_cast<panda_error>(hipLaunchHostFunc(static_cast<hipStream_t>(stream.handle), fn, user_data));
}

panda_error panda_event_create(panda_event *event, bool blocking_sync, bool disable_timing)
{
    unsigned flags = (blocking_sync ? hipEventBlockingSync : hipEventDefault) | (disable_timing ? hipEventDisableTiming : hipEventDefault);
    hipEvent_t e = nullptr;
    hipError_t err = hipEventCreateWithFlags(&e, flags);
    event->handle = e;
    return static_cast<panda_error>(err);
}

panda_error panda_event_record(panda_event event, panda_stream stream)
{
    return static_cast<panda_error>(hipEventRecord(static_cast<hipEvent_t>(event.handle), static_cast<hipStream_t>(stream.handle)));
}

panda_error panda_event_sync(panda_event event) { return static_cast<panda_error>(hipEventSynchronize(static_cast<hipEvent_t>(event.handle))); }

panda_error panda_event_query(panda_event event) { return static_cast<panda_error>(hipEventQuery(static_cast<hipEvent_t>(event.handle))); }

panda_error panda_event_destroy(panda_event event) { return static_cast<panda_error>(hipEventDestroy(static_cast<hipEvent_t>(event.handle))); }

panda_error panda_mem_get_info(size_t *free, size_t *total) { return static_cast<panda_error>(hipMemGetInfo(free, total)); }

panda_error panda_malloc(void **ptr, size_t size)
{
    const hipError_t e = hipMalloc(ptr, size);
    if (e == hipSuccess && ptr) panda::track_allocation(*ptr, size);
    return static_cast<panda_error>(e);
}

panda_error panda_malloc_host(void **ptr, size_t size) { return static_cast<panda_error>(hipHostMalloc(ptr, size, hipHostMallocDefault)); }

panda_error panda_free(void *ptr)
{
    panda::registry_forget_allocation(ptr); // a cached-bases registration must not survive its buffer (msm.hip)
    panda::untrack_allocation(ptr);
    return static_cast<panda_error>(hipFree(ptr));
}

panda_error panda_free_host(void *ptr) { return static_cast<panda_error>(hipHostFree(ptr)); }

panda_error panda_host_register(void *ptr, size_t size) { return static_cast<panda_error>(hipHostRegister(ptr, size, hipHostRegisterDefault)); }

panda_error panda_host_unregister(void *ptr) { return static_cast<panda_error>(hipHostUnregister(ptr)); }

panda_error panda_memcpy(void *dst, const void *src, size_t count) { return static_cast<panda_error>(hipMemcpy(dst, src, count, hipMemcpyDefault)); }

panda_error panda_memcpy_async(void *dst, const void *src, size_t count, panda_stream stream)
{
    return static_cast<panda_error>(hipMemcpyAsync(dst, src, count, hipMemcpyDefault, static_cast<hipStream_t>(stream.handle)));
}

panda_error panda_memset(void *ptr, int value, size_t count) { return static_cast<panda_error>(hipMemset(ptr, value, count)); }

panda_error panda_memset_async(void *ptr, int value, size_t count, panda_stream stream)
{
    return static_cast<panda_error>(hipMemsetAsync(ptr, value, count, static_cast<hipStream_t>(stream.handle)));
}

panda_error panda_mem_pool_create(panda_mem_pool *pool, int device_id)
{
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = device_id;
    hipMemPool_t p = nullptr;
    hipError_t e = hipMemPoolCreate(&p, &props);
    if (e != hipSuccess) {
        pool->handle = nullptr;
        return static_cast<panda_error>(e);
    }
    uint64_t threshold = UINT64_MAX; // keep freed blocks in the pool (common.cu:27-28)
    e = hipMemPoolSetAttribute(p, hipMemPoolAttrReleaseThreshold, &threshold);
    pool->handle = p;
    return static_cast<panda_error>(e);
}

panda_error panda_mem_pool_destroy(panda_mem_pool pool)
{
    panda::untrack_pool(pool.handle);
    return static_cast<panda_error>(hipMemPoolDestroy(static_cast<hipMemPool_t>(pool.handle)));
}

panda_error panda_malloc_from_pool_async(void **ptr, size_t size, panda_mem_pool pool, panda_stream stream)
{
    const hipError_t e = hipMallocFromPoolAsync(ptr, size, static_cast<hipMemPool_t>(pool.handle), static_cast<hipStream_t>(stream.handle));
    if (e == hipSuccess && ptr) panda::track_allocation(*ptr, size, pool.handle);
    return static_cast<panda_error>(e);
}

panda_error panda_free_async(void *ptr, panda_stream stream)
{
    panda::registry_forget_allocation(ptr);
    panda::untrack_allocation(ptr);
    return static_cast<panda_error>(hipFreeAsync(ptr, static_cast<hipStream_t>(stream.handle)));
}

panda_error panda_device_enable_peer_access(int device_id) { return static_cast<panda_error>(hipDeviceEnablePeerAccess(device_id, 0)); }

panda_error panda_device_disable_peer_access(int device_id) { return static_cast<panda_error>(hipDeviceDisablePeerAccess(device_id)); }

} // extern "C"
