// multi_gpu.hip -- the sharded MSM and NTT behind ONE C call, for a host that is not Python: one process, one host thread per
// device, RCCL over xGMI for the exchange ("one process, 8 devices, one ncclComm_t per device via
// ncclCommInitAll"; north_star: "Rust host code calling HIP through the existing C-ABI").  The reference has no counterpart -- it
// pins device 0 (src/cuda/core/unit/msm/msm_cuda.cuh:554-555, src/gpu_manager/wrapper.rs:38) and only declares the peer-access
// symbols (src/gpu_ffi/binding.rs:54-56).
//
//   MSM   shards by base-point range: device d runs the ordinary panda_msm_execute_* on its range (its own worker thread, so its own
//         scratch arena); the Jacobian partials (96 / 144 B each) are all-gathered with one ncclAllGather and added up on the host
//         (panda_msm_combine_*: EC addition is not an RCCL reduction operator).
//   NTT   shards by decimated slab: panda_ntt_slab_step1 on every device, ONE grouped ncclSend / ncclRecv all-to-all (chunk q of
//         rank d goes to rank q; every GPU pair has its own xGMI link, so this is not a ring), panda_ntt_slab_step2 on every device.
//         Everything is enqueued on the per-device streams; the call synchronises once at the end.
//
// Worker threads live as long as the handle: the library's scratch arenas and twiddle caches are per host thread, so a prover that
// repeats the same sharded call launches kernels only.
//
// PANDA_MULTI_LOOPBACK replaces RCCL by device-to-device copies and allows one device to play several ranks: the same threads,
// flags and chunk arithmetic as the RCCL path, testable on a one-GPU box (tests/test_gpu_parity.py, manager_test.cpp).
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "panda_internal.h"

namespace {

constexpr size_t MAX_RESULT_BYTES = 192; // BN254 G2 Jacobian

struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<hipError_t()> job;
    bool has_job = false, done = false, stop = false;
    hipError_t result = hipSuccess;
    int device = 0;

    void loop()
    {
        (void)hipSetDevice(device);
        for (;;) {
            std::function<hipError_t()> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [this] { return has_job || stop; });
                if (stop) break;
                j = std::move(job);
                has_job = false;
            }
            hipError_t e = j();
            {
                std::lock_guard<std::mutex> lk(m);
                result = e;
                done = true;
            }
            cv.notify_all();
        }
        // the thread's scratch arena and twiddle caches are released by their thread_local destructors
    }
    void submit(std::function<hipError_t()> j)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            has_job = true;
            done = false;
        }
        cv.notify_all();
    }
    hipError_t wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return done; });
        return result;
    }
};

struct MultiGpu {
    unsigned n = 0;
    unsigned transport = PANDA_MULTI_RCCL;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;                 // used when a configuration carries no stream of its own
    std::vector<hipStream_t> copy_streams;            // per device: the upload stream of the *_from_host_multi calls
    std::vector<hipStream_t> xchg_streams;            // per device: the exchange stream of the batched transforms (ntt_multi_batch)
    std::vector<std::vector<hipEvent_t>> events;      // per device: event pool of the batched transforms (grown on demand, kept)
    std::vector<void *> d_gather;                     // per device: n x MAX_RESULT_BYTES, partials of every rank
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<float> phase_ms;                      // n x PANDA_MSM_PHASES, from the workers' last MSM
    std::mutex call_mutex;                            // one sharded call at a time per handle

    template <class Fn>
    hipError_t on_all(Fn fn) // fn(rank) on every worker, with its device current; first error wins
    {
        for (unsigned d = 0; d < n; d++) workers[d]->submit([fn, d] { return fn(d); });
        hipError_t first = hipSuccess;
        for (unsigned d = 0; d < n; d++) {
            hipError_t e = workers[d]->wait();
            if (first == hipSuccess) first = e;
        }
        return first;
    }
    ~MultiGpu()
    {
        for (auto &w : workers) {
            if (!w) continue;
            {
                std::lock_guard<std::mutex> lk(w->m);
                w->stop = true;
            }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
        for (unsigned d = 0; d < comms.size(); d++)
            if (comms[d]) (void)ncclCommDestroy(comms[d]);
        for (unsigned d = 0; d < n && d < devices.size(); d++) {
            (void)hipSetDevice(devices[d]);
            if (d < streams.size() && streams[d]) (void)hipStreamDestroy(streams[d]);
            if (d < copy_streams.size() && copy_streams[d]) (void)hipStreamDestroy(copy_streams[d]);
            if (d < xchg_streams.size() && xchg_streams[d]) (void)hipStreamDestroy(xchg_streams[d]);
            if (d < events.size())
                for (hipEvent_t ev : events[d]) (void)hipEventDestroy(ev);
            if (d < d_gather.size() && d_gather[d]) (void)hipFree(d_gather[d]);
        }
    }
};

hipError_t from_nccl(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return hipSuccess;
    fprintf(stderr, "[panda-hip] RCCL error in %s: %s\n", what, ncclGetErrorString(r));
    return hipErrorUnknown;
}
#define PANDA_TRY_NCCL(expr) PANDA_TRY(from_nccl((expr), #expr))

hipStream_t stream_of(const MultiGpu &mg, unsigned d, panda_stream s) { return s.handle ? static_cast<hipStream_t>(s.handle) : mg.streams[d]; }

hipError_t sync_all(MultiGpu &mg, const std::vector<hipStream_t> &streams)
{
    for (unsigned d = 0; d < mg.n; d++) {
        PANDA_TRY(hipSetDevice(mg.devices[d]));
        PANDA_TRY(hipStreamSynchronize(streams[d]));
    }
    return hipSuccess;
}

typedef panda_error (*combine_fn)(const void *, unsigned, panda_msm_result_coordinate_type, void *);

// an RCCL group that is closed on every path: a failing call between ncclGroupStart and ncclGroupEnd must not leave the thread's group
// open (the next collective of this host thread would be queued into the stale group and could hang)
template <class Body>
hipError_t nccl_group(Body body)
{
    PANDA_TRY_NCCL(ncclGroupStart());
    const hipError_t e = body();
    const ncclResult_t end = ncclGroupEnd();
    if (e != hipSuccess) return e;
    return from_nccl(end, "ncclGroupEnd");
}

// `execute(d, cfg)`: the single-GPU call of rank d (synchronous), run on worker d with its device current
hipError_t msm_multi(MultiGpu &mg, const panda_msm_configuration *cfgs, void *result, size_t rb, const std::function<panda_error(unsigned, const panda_msm_configuration &)> &execute,
                     combine_fn combine)
{
    if (!cfgs || !result) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> call(mg.call_mutex);
    int caller_dev = 0;
    PANDA_TRY(hipGetDevice(&caller_dev));
    std::vector<hipStream_t> streams(mg.n);
    for (unsigned d = 0; d < mg.n; d++) streams[d] = stream_of(mg, d, cfgs[d].stream);
    // every device: the ordinary single-GPU call on its base range (synchronous), then its partial into slot d of its gather buffer
    hipError_t e = mg.on_all([&](unsigned d) -> hipError_t {
        panda_msm_configuration c = cfgs[d];
        c.msm_result_coordinate_type = JACOBIAN; // partials are added as Jacobian points; the requested form is produced by the combine
        c.stream.handle = streams[d];
        const panda_error pe = execute(d, c);
        if (pe != panda_success) return static_cast<hipError_t>(pe);
        (void)panda_msm_last_phase_ms(&mg.phase_ms[d * PANDA_MSM_PHASES]);
        return hipMemcpyAsync((char *)mg.d_gather[d] + d * rb, c.results, rb, hipMemcpyDefault, streams[d]);
    });
    if (e == hipSuccess) {
        if (mg.transport == PANDA_MULTI_RCCL) {
            // one collective: every device ends with all partials (in place: the send buffer is the rank's own slot)
            e = nccl_group([&]() -> hipError_t {
                for (unsigned d = 0; d < mg.n; d++)
                    PANDA_TRY_NCCL(ncclAllGather((char *)mg.d_gather[d] + d * rb, mg.d_gather[d], rb, ncclChar, mg.comms[d], streams[d]));
                return hipSuccess;
            });
        } else {
            e = sync_all(mg, streams);
            for (unsigned d = 0; e == hipSuccess && d < mg.n; d++) {
                e = hipSetDevice(mg.devices[d]);
                for (unsigned q = 0; e == hipSuccess && q < mg.n; q++)
                    if (q != d) e = hipMemcpyAsync((char *)mg.d_gather[d] + q * rb, (char *)mg.d_gather[q] + q * rb, rb, hipMemcpyDefault, streams[d]);
            }
        }
    }
    // on every path, errors included: nothing of this call is in flight on the callers' streams when it returns
    const hipError_t drained = sync_all(mg, streams);
    if (e == hipSuccess) e = drained;
    if (e == hipSuccess) {
        std::vector<unsigned char> partials(mg.n * rb);
        e = hipSetDevice(mg.devices[0]);
        if (e == hipSuccess) e = hipMemcpy(partials.data(), mg.d_gather[0], mg.n * rb, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = static_cast<hipError_t>(combine(partials.data(), mg.n, cfgs[0].msm_result_coordinate_type, result));
    }
    (void)hipSetDevice(caller_dev);
    return e;
}

typedef panda_error (*slab_fn)(const panda_ntt_slab_configuration);

// forward: step1 -> all-to-all -> step2; inverse: inverse_step1 -> the same all-to-all -> inverse_step2
hipError_t ntt_multi(MultiGpu &mg, const panda_ntt_slab_configuration *cfgs, slab_fn first, slab_fn second)
{
    if (!cfgs) return hipErrorInvalidValue;
    unsigned log_ranks = 0;
    while ((1u << log_ranks) < mg.n) log_ranks++;
    if ((1u << log_ranks) != mg.n) return hipErrorInvalidValue; // the slab decomposition wants a power of two
    for (unsigned d = 0; d < mg.n; d++)
        if (cfgs[d].rank != d || cfgs[d].log_ranks != log_ranks || cfgs[d].log_n != cfgs[0].log_n || cfgs[d].log_n < 2 * log_ranks || !cfgs[d].d_slab ||
            !cfgs[d].d_scratch || !cfgs[d].omega)
            return hipErrorInvalidValue;
    std::lock_guard<std::mutex> call(mg.call_mutex);
    int caller_dev = 0;
    PANDA_TRY(hipGetDevice(&caller_dev));
    const size_t slab_bytes = (size_t)32 << (cfgs[0].log_n - log_ranks), chunk = slab_bytes >> log_ranks;
    std::vector<hipStream_t> streams(mg.n);
    std::vector<unsigned> flag(mg.n, 0);
    std::vector<char *> src(mg.n), dst(mg.n);
    for (unsigned d = 0; d < mg.n; d++) streams[d] = stream_of(mg, d, cfgs[d].stream);
    hipError_t e = mg.on_all([&](unsigned d) -> hipError_t {
        panda_ntt_slab_configuration c = cfgs[d];
        c.stream.handle = streams[d];
        c.flag = &flag[d];
        return static_cast<hipError_t>(first(c)); // enqueued; the flag is valid on return
    });
    if (e == hipSuccess) {
        for (unsigned d = 0; d < mg.n; d++) {
            src[d] = (char *)(flag[d] ? cfgs[d].d_scratch : cfgs[d].d_slab);
            dst[d] = (char *)(flag[d] ? cfgs[d].d_slab : cfgs[d].d_scratch);
        }
        if (mg.transport == PANDA_MULTI_RCCL) {
            // the single exchange: chunk q of rank d -> chunk d of rank q, all pairs in one group (full mesh, one xGMI link per pair)
            e = nccl_group([&]() -> hipError_t {
                for (unsigned d = 0; d < mg.n; d++)
                    for (unsigned q = 0; q < mg.n; q++) {
                        PANDA_TRY_NCCL(ncclSend(src[d] + q * chunk, chunk, ncclChar, (int)q, mg.comms[d], streams[d]));
                        PANDA_TRY_NCCL(ncclRecv(dst[d] + q * chunk, chunk, ncclChar, (int)q, mg.comms[d], streams[d]));
                    }
                return hipSuccess;
            });
        } else {
            e = sync_all(mg, streams); // every rank's step 1 is complete before anybody copies out of it
            for (unsigned d = 0; e == hipSuccess && d < mg.n; d++) {
                e = hipSetDevice(mg.devices[d]);
                for (unsigned q = 0; e == hipSuccess && q < mg.n; q++)
                    e = hipMemcpyAsync(dst[d] + q * chunk, src[q] + d * chunk, chunk, hipMemcpyDefault, streams[d]);
            }
        }
    }
    if (e == hipSuccess)
        e = mg.on_all([&](unsigned d) -> hipError_t {
            panda_ntt_slab_configuration c = cfgs[d];
            c.stream.handle = streams[d];
            c.d_slab = dst[d];
            c.d_scratch = src[d];
            c.flag = &flag[d];
            const panda_error pe = second(c);
            if (pe != panda_success) return static_cast<hipError_t>(pe);
            return hipStreamSynchronize(streams[d]);
        });
    if (e != hipSuccess) (void)sync_all(mg, streams); // nothing of a failed call stays in flight on the callers' streams
    if (e == hipSuccess)
        for (unsigned d = 0; d < mg.n; d++) {
            const char *out = flag[d] ? src[d] : dst[d];
            if (cfgs[d].flag) *(unsigned *)cfgs[d].flag = out == (const char *)cfgs[d].d_scratch ? 1u : 0u;
        }
    (void)hipSetDevice(caller_dev);
    return e;
}

// A batch of sharded transforms with the exchange of transform t hidden behind the kernels of its neighbours ("a sharded
// transform that overlaps its exchange").  Inside ONE transform the all-to-all cannot start before the last pass of step 1 has finished (every tile
// of that pass writes into every rank's chunk); a prover, however, transforms many polynomials with the same root, so the
// batch is pipelined instead: every device has a compute stream (the caller's, or the handle's) and an exchange stream,
//     compute  : step1(0) step1(1) step2(0) step1(2) step2(1) ...          exchange : x(0) x(1) x(2) ...
// with x(t) waiting for step1(t) by event and step2(t) waiting for x(t).  Transform t's slab and scratch are its own (cfgs[t * n + d]); layouts, flags
// and results are those of `count` separate panda_ntt_execute_*_multi calls.  One host synchronisation at the end.
hipError_t ntt_multi_batch(MultiGpu &mg, const panda_ntt_slab_configuration *cfgs, unsigned count, slab_fn first, slab_fn second)
{
    if (!cfgs || count == 0 || count > 4096) return hipErrorInvalidValue;
    unsigned log_ranks = 0;
    while ((1u << log_ranks) < mg.n) log_ranks++;
    if ((1u << log_ranks) != mg.n) return hipErrorInvalidValue;
    const unsigned n = mg.n;
    for (unsigned t = 0; t < count; t++)
        for (unsigned d = 0; d < n; d++) {
            const panda_ntt_slab_configuration &c = cfgs[(size_t)t * n + d];
            if (c.rank != d || c.log_ranks != log_ranks || c.log_n != cfgs[0].log_n || c.log_n < 2 * log_ranks || !c.d_slab || !c.d_scratch || !c.omega) return hipErrorInvalidValue;
        }
    std::lock_guard<std::mutex> call(mg.call_mutex);
    int caller_dev = 0;
    PANDA_TRY(hipGetDevice(&caller_dev));
    const size_t slab_bytes = (size_t)32 << (cfgs[0].log_n - log_ranks), chunk = slab_bytes >> log_ranks;
    std::vector<hipStream_t> streams(n);
    for (unsigned d = 0; d < n; d++) streams[d] = stream_of(mg, d, cfgs[d].stream);
    // two events per transform and device: step 1 enqueued-and-done (E1), exchange done (E2)
    hipError_t e = hipSuccess;
    for (unsigned d = 0; e == hipSuccess && d < n; d++) {
        e = hipSetDevice(mg.devices[d]);
        while (e == hipSuccess && mg.events[d].size() < 2 * (size_t)count) {
            hipEvent_t ev = nullptr;
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) mg.events[d].push_back(ev);
        }
    }
    auto E1 = [&](unsigned t, unsigned d) { return mg.events[d][2 * t]; };
    auto E2 = [&](unsigned t, unsigned d) { return mg.events[d][2 * t + 1]; };
    std::vector<unsigned> flag1((size_t)count * n, 0), flag2((size_t)count * n, 0);
    std::vector<char *> src((size_t)count * n), dst((size_t)count * n);
    for (unsigned t = 0; e == hipSuccess && t <= count; t++) {
        // compute streams: step 1 of transform t, then step 2 of transform t - 1 (whose exchange is under way or done)
        e = mg.on_all([&, t](unsigned d) -> hipError_t {
            if (t < count) {
                panda_ntt_slab_configuration c = cfgs[(size_t)t * n + d];
                c.stream.handle = streams[d];
                c.flag = &flag1[(size_t)t * n + d];
                const panda_error pe = first(c); // enqueued; the flag is valid on return
                if (pe != panda_success) return static_cast<hipError_t>(pe);
                PANDA_TRY(hipEventRecord(E1(t, d), streams[d]));
            }
            if (t > 0) {
                const unsigned u = t - 1;
                // RCCL: my receives are complete when my exchange stream reaches E2.  Loopback: the peers copy OUT of my step-1 buffer on their
                // exchange streams, and step 2 uses that buffer as its scratch: wait for all of them.
                if (mg.transport == PANDA_MULTI_RCCL)
                    PANDA_TRY(hipStreamWaitEvent(streams[d], E2(u, d), 0));
                else
                    for (unsigned q = 0; q < n; q++) PANDA_TRY(hipStreamWaitEvent(streams[d], E2(u, q), 0));
                panda_ntt_slab_configuration c = cfgs[(size_t)u * n + d];
                c.stream.handle = streams[d];
                c.d_slab = dst[(size_t)u * n + d];
                c.d_scratch = src[(size_t)u * n + d];
                c.flag = &flag2[(size_t)u * n + d];
                const panda_error pe = second(c);
                if (pe != panda_success) return static_cast<hipError_t>(pe);
            }
            return hipSuccess;
        });
        if (e != hipSuccess || t == count) break;
        // exchange streams: transform t's all-to-all behind its step 1
        for (unsigned d = 0; d < n; d++) {
            const panda_ntt_slab_configuration &c = cfgs[(size_t)t * n + d];
            const bool f = flag1[(size_t)t * n + d] != 0;
            src[(size_t)t * n + d] = (char *)(f ? c.d_scratch : c.d_slab);
            dst[(size_t)t * n + d] = (char *)(f ? c.d_slab : c.d_scratch);
        }
        char **sr = &src[(size_t)t * n], **ds = &dst[(size_t)t * n];
        if (mg.transport == PANDA_MULTI_RCCL) {
            for (unsigned d = 0; e == hipSuccess && d < n; d++) {
                e = hipSetDevice(mg.devices[d]);
                if (e == hipSuccess) e = hipStreamWaitEvent(mg.xchg_streams[d], E1(t, d), 0);
            }
            if (e == hipSuccess)
                e = nccl_group([&]() -> hipError_t {
                    for (unsigned d = 0; d < n; d++)
                        for (unsigned q = 0; q < n; q++) {
                            PANDA_TRY_NCCL(ncclSend(sr[d] + q * chunk, chunk, ncclChar, (int)q, mg.comms[d], mg.xchg_streams[d]));
                            PANDA_TRY_NCCL(ncclRecv(ds[d] + q * chunk, chunk, ncclChar, (int)q, mg.comms[d], mg.xchg_streams[d]));
                        }
                    return hipSuccess;
                });
        } else {
            for (unsigned d = 0; e == hipSuccess && d < n; d++) {
                e = hipSetDevice(mg.devices[d]);
                for (unsigned q = 0; e == hipSuccess && q < n; q++) e = hipStreamWaitEvent(mg.xchg_streams[d], E1(t, q), 0); // every rank's step 1
                for (unsigned q = 0; e == hipSuccess && q < n; q++)
                    e = hipMemcpyAsync(ds[d] + q * chunk, sr[q] + d * chunk, chunk, hipMemcpyDefault, mg.xchg_streams[d]);
            }
        }
        for (unsigned d = 0; e == hipSuccess && d < n; d++) {
            e = hipSetDevice(mg.devices[d]);
            if (e == hipSuccess) e = hipEventRecord(E2(t, d), mg.xchg_streams[d]);
        }
    }
    // one wait at the end; on errors too nothing of this call stays in flight
    const hipError_t drained_x = sync_all(mg, mg.xchg_streams), drained_c = sync_all(mg, streams);
    if (e == hipSuccess) e = drained_x;
    if (e == hipSuccess) e = drained_c;
    if (e == hipSuccess)
        for (unsigned t = 0; t < count; t++)
            for (unsigned d = 0; d < n; d++) {
                const panda_ntt_slab_configuration &c = cfgs[(size_t)t * n + d];
                const size_t i = (size_t)t * n + d;
                const char *out = flag2[i] ? src[i] : dst[i];
                if (c.flag) *(unsigned *)c.flag = out == (const char *)c.d_scratch ? 1u : 0u;
            }
    (void)hipSetDevice(caller_dev);
    return e;
}

MultiGpu *handle_of(panda_multi_gpu mg) { return static_cast<MultiGpu *>(mg.handle); }

} // namespace

extern "C" {

panda_error panda_multi_gpu_create(panda_multi_gpu *out, const int *devices, unsigned n_dev, unsigned transport)
{
    if (!out || !devices || n_dev == 0 || n_dev > 64 || transport > PANDA_MULTI_LOOPBACK) return panda_error_invalid_value;
    int count = 0, caller_dev = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || hipGetDevice(&caller_dev) != hipSuccess) return panda_error_invalid_value;
    // Test-only knob: with PANDA_TEST_SHARED_DEVICE_RCCL set, several RCCL ranks may name one device.  Real RCCL refuses that itself; the knob
    // exists for tests/fake_rccl (an LD_PRELOAD stand-in that validates and performs the RCCL calls of this file on a one-GPU box).
    const bool one_rank_per_device = transport == PANDA_MULTI_RCCL && !getenv("PANDA_TEST_SHARED_DEVICE_RCCL");
    for (unsigned d = 0; d < n_dev; d++) {
        if (devices[d] < 0 || devices[d] >= count) return panda_error_invalid_value;
        if (one_rank_per_device)
            for (unsigned q = 0; q < d; q++)
                if (devices[q] == devices[d]) return panda_error_invalid_value; // one RCCL rank per device
    }
    std::unique_ptr<MultiGpu> mg(new MultiGpu());
    mg->n = n_dev;
    mg->transport = transport;
    mg->devices.assign(devices, devices + n_dev);
    mg->streams.assign(n_dev, nullptr);
    mg->copy_streams.assign(n_dev, nullptr);
    mg->xchg_streams.assign(n_dev, nullptr);
    mg->events.assign(n_dev, std::vector<hipEvent_t>());
    mg->d_gather.assign(n_dev, nullptr);
    mg->phase_ms.assign((size_t)n_dev * PANDA_MSM_PHASES, 0.f);
    hipError_t e = hipSuccess;
    for (unsigned d = 0; e == hipSuccess && d < n_dev; d++) {
        e = hipSetDevice(devices[d]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&mg->streams[d], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&mg->copy_streams[d], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&mg->xchg_streams[d], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&mg->d_gather[d], (size_t)n_dev * MAX_RESULT_BYTES);
    }
    if (e == hipSuccess && transport == PANDA_MULTI_RCCL) {
        mg->comms.assign(n_dev, nullptr);
        e = from_nccl(ncclCommInitAll(mg->comms.data(), (int)n_dev, devices), "ncclCommInitAll");
    }
    (void)hipSetDevice(caller_dev);
    if (e != hipSuccess) return static_cast<panda_error>(e);
    for (unsigned d = 0; d < n_dev; d++) {
        mg->workers.emplace_back(new Worker());
        Worker *w = mg->workers.back().get();
        w->device = devices[d];
        w->th = std::thread([w] { w->loop(); });
    }
    out->handle = mg.release();
    return panda_success;
}

panda_error panda_multi_gpu_destroy(panda_multi_gpu mg)
{
    if (!mg.handle) return panda_error_invalid_value;
    int caller_dev = 0;
    (void)hipGetDevice(&caller_dev);
    delete handle_of(mg);
    (void)hipSetDevice(caller_dev);
    return panda_success;
}

panda_error panda_multi_gpu_device_count(panda_multi_gpu mg, unsigned *n_dev)
{
    if (!mg.handle || !n_dev) return panda_error_invalid_value;
    *n_dev = handle_of(mg)->n;
    return panda_success;
}

panda_error panda_msm_execute_bn254_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, void *result)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        msm_multi(*handle_of(mg), cfgs, result, 96, [](unsigned, const panda_msm_configuration &c) { return panda_msm_execute_bn254(c); }, panda_msm_combine_bn254));
}

panda_error panda_msm_execute_bls12_377_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, void *result)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        msm_multi(*handle_of(mg), cfgs, result, 144, [](unsigned, const panda_msm_configuration &c) { return panda_msm_execute_bls12_377(c); }, panda_msm_combine_bls12_377));
}

panda_error panda_msm_execute_bls12_381_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, void *result)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        msm_multi(*handle_of(mg), cfgs, result, 144, [](unsigned, const panda_msm_configuration &c) { return panda_msm_execute_bls12_381(c); }, panda_msm_combine_bls12_381));
}

panda_error panda_msm_execute_bn254_g2_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, void *result)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        msm_multi(*handle_of(mg), cfgs, result, 192, [](unsigned, const panda_msm_configuration &c) { return panda_msm_execute_bn254_g2(c); }, panda_msm_combine_bn254_g2));
}

// Scalars that start on the HOST (north_star / SURVEY 8e: "scalars H2D'd per shard"; unit.rs:103-188 stages them before it executes):
// every worker runs the in-call upload pipeline of panda_msm_execute_from_host on its own shard -- its own arena, helper stream, copy
// stream and PCIe link -- so the G uploads run side by side and each hides behind its shard's kernels.
static panda_error msm_from_host_multi(panda_multi_gpu mg, unsigned curve, size_t rb, combine_fn combine, const panda_msm_configuration *cfgs, const void *const *h_scalars,
                                       unsigned ranges, void *result)
{
    if (!mg.handle || !h_scalars) return panda_error_invalid_value;
    MultiGpu &m = *handle_of(mg);
    for (unsigned d = 0; d < m.n; d++)
        if (!h_scalars[d]) return panda_error_invalid_value;
    return static_cast<panda_error>(msm_multi(
        m, cfgs, result, rb,
        [&m, curve, h_scalars, ranges](unsigned d, const panda_msm_configuration &c) {
            panda_stream h2d;
            h2d.handle = m.copy_streams[d];
            return panda_msm_execute_from_host(curve, c, h_scalars[d], ranges, h2d);
        },
        combine));
}

panda_error panda_msm_execute_bn254_from_host_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, const void *const *h_scalars, unsigned ranges, void *result)
{
    return msm_from_host_multi(mg, 0, 96, panda_msm_combine_bn254, cfgs, h_scalars, ranges, result);
}

panda_error panda_msm_execute_bls12_377_from_host_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, const void *const *h_scalars, unsigned ranges,
                                                        void *result)
{
    return msm_from_host_multi(mg, 1, 144, panda_msm_combine_bls12_377, cfgs, h_scalars, ranges, result);
}

panda_error panda_msm_execute_bls12_381_from_host_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, const void *const *h_scalars, unsigned ranges,
                                                        void *result)
{
    return msm_from_host_multi(mg, 2, 144, panda_msm_combine_bls12_381, cfgs, h_scalars, ranges, result);
}

panda_error panda_msm_execute_bn254_g2_from_host_multi(panda_multi_gpu mg, const panda_msm_configuration *cfgs, const void *const *h_scalars, unsigned ranges, void *result)
{
    return msm_from_host_multi(mg, 3, 192, panda_msm_combine_bn254_g2, cfgs, h_scalars, ranges, result);
}

panda_error panda_ntt_execute_bn254_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(ntt_multi(*handle_of(mg), cfgs, panda_ntt_slab_step1_bn254_enqueue, panda_ntt_slab_step2_bn254_enqueue));
}

panda_error panda_ntt_execute_bn254_inverse_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        ntt_multi(*handle_of(mg), cfgs, panda_ntt_slab_inverse_step1_bn254_enqueue, panda_ntt_slab_inverse_step2_bn254_enqueue));
}

panda_error panda_ntt_execute_bn254_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs, unsigned count)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(ntt_multi_batch(*handle_of(mg), cfgs, count, panda_ntt_slab_step1_bn254_enqueue, panda_ntt_slab_step2_bn254_enqueue));
}

panda_error panda_ntt_execute_bn254_inverse_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs, unsigned count)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        ntt_multi_batch(*handle_of(mg), cfgs, count, panda_ntt_slab_inverse_step1_bn254_enqueue, panda_ntt_slab_inverse_step2_bn254_enqueue));
}

// the same sharded transforms over the BLS12-377 scalar field
panda_error panda_ntt_execute_bls12_377_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(ntt_multi(*handle_of(mg), cfgs, panda_ntt_slab_step1_bls12_377_enqueue, panda_ntt_slab_step2_bls12_377_enqueue));
}

panda_error panda_ntt_execute_bls12_377_inverse_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        ntt_multi(*handle_of(mg), cfgs, panda_ntt_slab_inverse_step1_bls12_377_enqueue, panda_ntt_slab_inverse_step2_bls12_377_enqueue));
}

panda_error panda_ntt_execute_bls12_377_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs, unsigned count)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        ntt_multi_batch(*handle_of(mg), cfgs, count, panda_ntt_slab_step1_bls12_377_enqueue, panda_ntt_slab_step2_bls12_377_enqueue));
}

panda_error panda_ntt_execute_bls12_377_inverse_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs, unsigned count)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        ntt_multi_batch(*handle_of(mg), cfgs, count, panda_ntt_slab_inverse_step1_bls12_377_enqueue, panda_ntt_slab_inverse_step2_bls12_377_enqueue));
}

// the same sharded transforms over the BLS12-381 scalar field
panda_error panda_ntt_execute_bls12_381_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(ntt_multi(*handle_of(mg), cfgs, panda_ntt_slab_step1_bls12_381_enqueue, panda_ntt_slab_step2_bls12_381_enqueue));
}

panda_error panda_ntt_execute_bls12_381_inverse_multi(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        ntt_multi(*handle_of(mg), cfgs, panda_ntt_slab_inverse_step1_bls12_381_enqueue, panda_ntt_slab_inverse_step2_bls12_381_enqueue));
}

panda_error panda_ntt_execute_bls12_381_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs, unsigned count)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        ntt_multi_batch(*handle_of(mg), cfgs, count, panda_ntt_slab_step1_bls12_381_enqueue, panda_ntt_slab_step2_bls12_381_enqueue));
}

panda_error panda_ntt_execute_bls12_381_inverse_multi_batch(panda_multi_gpu mg, const panda_ntt_slab_configuration *cfgs, unsigned count)
{
    if (!mg.handle) return panda_error_invalid_value;
    return static_cast<panda_error>(
        ntt_multi_batch(*handle_of(mg), cfgs, count, panda_ntt_slab_inverse_step1_bls12_381_enqueue, panda_ntt_slab_inverse_step2_bls12_381_enqueue));
}

panda_error panda_multi_gpu_last_phase_ms(panda_multi_gpu mg, unsigned rank, float *ms)
{
    if (!mg.handle || !ms || rank >= handle_of(mg)->n) return panda_error_invalid_value;
    for (int i = 0; i < PANDA_MSM_PHASES; i++) ms[i] = handle_of(mg)->phase_ms[(size_t)rank * PANDA_MSM_PHASES + i];
    return panda_success;
}

} // extern "C"
