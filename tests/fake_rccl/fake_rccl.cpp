// fake_rccl.cpp -- TEST INFRASTRUCTURE, not product code.  A stand-in for the eight RCCL entry points csrc/multi_gpu.hip calls
// (ncclCommInitAll, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclAllGather, ncclSend, ncclRecv, ncclGetErrorString), loaded with
// LD_PRELOAD into a fresh child process so that the PANDA_MULTI_RCCL transport can run with 2 / 4 / 8 ranks on a box that has ONE GPU
// (several ranks share device 0).  The product library neither links nor knows this file: it keeps its -lrccl and the dynamic linker
// binds its ncclXxx references here because a preloaded object comes first in the lookup order.
//
// What it does with every call:
//   (a) VALIDATES the arguments the way RCCL's single-process multi-communicator mode needs them:
//         * the communicator is alive and belongs to a clique made by ncclCommInitAll;
//         * the stream lives on the communicator's device; every buffer is device memory of that device and the whole transfer
//           lies inside its allocation;
//         * a collective or a point-to-point call on a clique of more than one rank is made inside ncclGroupStart / ncclGroupEnd
//           (outside a group one host thread would block for ever in real RCCL);
//         * one stream per communicator per group;
//         * ncclAllGather: every rank of the clique calls it, with the same count and type; a send buffer that overlaps the receive
//           buffer must be exactly recvbuff + rank * count * size (the in-place rule);
//         * ncclSend / ncclRecv: the k-th send of rank d to peer q is matched with the k-th receive of rank q from peer d, inside the
//           same group, with equal counts and types; nothing may be left unmatched at ncclGroupEnd; peers are inside the clique.
//   (b) PERFORMS the transfer with RCCL's completion semantics: the group starts on every participating stream after the work
//       already queued there (sender's stream records an event, receiver's stream waits for it), the copies run on the RECEIVER's
//       stream, and a sender's stream does not pass the group before its readers are done (receiver records, sender waits).
//
// A failed validation is counted, written to the log and to stderr ("[fake-rccl] FAIL ...") and returned as ncclInvalidUsage /
// ncclInvalidArgument.  FAKE_RCCL_LOG names a file that receives one line per communicator clique, group and failure;
// fake_rccl_counters() hands the totals to the test.  The signatures come from <rccl/rccl.h> itself, so a drift is a compile error.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr uint32_t LIVE = 0xFA4ECC11u, DEAD = 0xDEADCC11u;

struct Clique;
struct FakeComm {
    uint32_t magic = LIVE;
    int rank = 0, nranks = 0, device = 0;
    Clique *clique = nullptr;
    hipEvent_t ready = nullptr, done = nullptr; // "everything queued before the group has run" / "my copies of this group are done"
};
struct Clique {
    unsigned id = 0;
    std::vector<FakeComm *> comms;
    int alive = 0;
};

enum Kind { ALLGATHER, SEND, RECV };
struct Op {
    Kind kind;
    FakeComm *comm;
    const void *sendbuff;
    void *recvbuff;
    size_t count;
    ncclDataType_t type;
    int peer;
    hipStream_t stream;
};

struct Counters {
    uint64_t cliques = 0, groups = 0, allgathers = 0, allgather_ranks = 0, exchanges = 0, matched_pairs = 0, bytes = 0, failures = 0, calls_outside_group = 0,
             largest_clique = 0;
};

std::mutex g_mutex; // cliques, counters, the log
Counters g_cnt;
unsigned g_next_clique = 1;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

void logf(const char *fmt, ...)
{
    char line[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    const char *path = getenv("FAKE_RCCL_LOG");
    if (path && *path) {
        if (FILE *f = fopen(path, "a")) {
            fprintf(f, "%s\n", line);
            fclose(f);
        }
    }
    if (strncmp(line, "FAIL", 4) == 0 || getenv("FAKE_RCCL_VERBOSE")) fprintf(stderr, "[fake-rccl] %s\n", line);
}

#define FAIL(code, ...)          \
    do {                         \
        {                        \
            std::lock_guard<std::mutex> lk(g_mutex); \
            g_cnt.failures++;    \
        }                        \
        logf("FAIL " __VA_ARGS__); \
        return code;             \
    } while (0)

size_t size_of(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8:
    case ncclUint8:
    case ncclFloat8e4m3:
    case ncclFloat8e5m2:
        return 1;
    case ncclFloat16:
    case ncclBfloat16:
        return 2;
    case ncclInt32:
    case ncclUint32:
    case ncclFloat32:
        return 4;
    case ncclInt64:
    case ncclUint64:
    case ncclFloat64:
        return 8;
    default:
        return 0;
    }
}

FakeComm *comm_of(ncclComm_t c)
{
    FakeComm *fc = reinterpret_cast<FakeComm *>(c);
    return fc && fc->magic == LIVE ? fc : nullptr;
}

// device memory of `device`, with [p, p + bytes) inside one allocation
ncclResult_t check_buffer(const void *p, size_t bytes, int device, const char *what, int rank)
{
    if (!p) FAIL(ncclInvalidArgument, "%s of rank %d is a null pointer", what, rank);
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof attr);
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        FAIL(ncclInvalidArgument, "%s of rank %d (%p) is not memory the runtime knows", what, rank, p);
    }
    if (attr.type != hipMemoryTypeDevice) FAIL(ncclInvalidArgument, "%s of rank %d (%p) is not device memory (type %d)", what, rank, p, (int)attr.type);
    if (attr.device != device) FAIL(ncclInvalidArgument, "%s of rank %d (%p) lives on device %d, the communicator on device %d", what, rank, p, attr.device, device);
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
        (void)hipGetLastError();
        FAIL(ncclInvalidArgument, "%s of rank %d (%p): no allocation found", what, rank, p);
    }
    if ((const char *)p + bytes > (const char *)base + size)
        FAIL(ncclInvalidArgument, "%s of rank %d: %zu bytes at %p run past the end of its allocation (%p + %zu)", what, rank, bytes, p, (void *)base, size);
    return ncclSuccess;
}

ncclResult_t check_stream(hipStream_t s, const FakeComm *c)
{
    if (!s) return ncclSuccess; // the null stream follows the current device; nothing to compare
    hipDevice_t dev = -1;
    if (hipStreamGetDevice(s, &dev) != hipSuccess) {
        (void)hipGetLastError();
        FAIL(ncclInvalidArgument, "the stream of rank %d (%p) is not a live stream", c->rank, (void *)s);
    }
    if ((int)dev != c->device) FAIL(ncclInvalidArgument, "the stream of rank %d lives on device %d, the communicator on device %d", c->rank, (int)dev, c->device);
    return ncclSuccess;
}

bool overlap(const void *a, size_t na, const void *b, size_t nb) { return (const char *)a < (const char *)b + nb && (const char *)b < (const char *)a + na; }

#define HIP_OK(expr)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) FAIL(ncclUnhandledCudaError, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct Transfer { // one copy on the receiver's stream
    FakeComm *from, *to;
    const void *src;
    void *dst;
    size_t bytes;
};

// the ops of one closed group: validate everything first, then enqueue
ncclResult_t run_group(std::vector<Op> &ops)
{
    if (ops.empty()) return ncclSuccess;
    // ---- per call
    std::map<FakeComm *, hipStream_t> stream_of;
    for (const Op &op : ops) {
        FakeComm *c = op.comm;
        if (c->magic != LIVE) FAIL(ncclInvalidArgument, "a communicator was destroyed while its call sat in an open group");
        const size_t es = size_of(op.type);
        if (es == 0) FAIL(ncclInvalidArgument, "rank %d: unknown data type %d", c->rank, (int)op.type);
        if (ncclResult_t r = check_stream(op.stream, c)) return r;
        auto it = stream_of.find(c);
        if (it == stream_of.end())
            stream_of[c] = op.stream;
        else if (it->second != op.stream)
            FAIL(ncclInvalidUsage, "rank %d uses two streams (%p, %p) inside one group", c->rank, (void *)it->second, (void *)op.stream);
        if (op.kind == ALLGATHER) {
            if (ncclResult_t r = check_buffer(op.sendbuff, op.count * es, c->device, "all-gather send buffer", c->rank)) return r;
            if (ncclResult_t r = check_buffer(op.recvbuff, op.count * es * c->nranks, c->device, "all-gather receive buffer", c->rank)) return r;
            const char *slot = (const char *)op.recvbuff + (size_t)c->rank * op.count * es;
            if (overlap(op.sendbuff, op.count * es, op.recvbuff, op.count * es * c->nranks) && (const char *)op.sendbuff != slot)
                FAIL(ncclInvalidArgument, "all-gather of rank %d: the send buffer %p overlaps the receive buffer %p but is not its slot %d (%p)", c->rank, op.sendbuff,
                     op.recvbuff, c->rank, (const void *)slot);
        } else {
            if (op.peer < 0 || op.peer >= c->nranks) FAIL(ncclInvalidArgument, "rank %d: peer %d is outside the clique of %d", c->rank, op.peer, c->nranks);
            const void *p = op.kind == SEND ? op.sendbuff : op.recvbuff;
            if (ncclResult_t r = check_buffer(p, op.count * es, c->device, op.kind == SEND ? "send buffer" : "receive buffer", c->rank)) return r;
        }
    }
    // ---- matching
    std::vector<Transfer> transfers;
    uint64_t n_allgathers = 0, n_ag_ranks = 0, n_pairs = 0, n_bytes = 0, n_exchange_cliques = 0;
    std::map<Clique *, std::vector<const Op *>> by_clique;
    for (const Op &op : ops) by_clique[op.comm->clique].push_back(&op);
    for (auto &kv : by_clique) {
        Clique *q = kv.first;
        const int n = (int)q->comms.size();
        // all-gathers: the k-th call of every rank belongs to the k-th collective
        std::vector<std::vector<const Op *>> ag(n);
        std::map<std::pair<int, int>, std::vector<const Op *>> sends, recvs; // (from, to)
        for (const Op *op : kv.second) {
            if (op->kind == ALLGATHER)
                ag[op->comm->rank].push_back(op);
            else if (op->kind == SEND)
                sends[{op->comm->rank, op->peer}].push_back(op);
            else
                recvs[{op->peer, op->comm->rank}].push_back(op);
        }
        size_t rounds = 0;
        for (int r = 0; r < n; r++) rounds = ag[r].size() > rounds ? ag[r].size() : rounds;
        for (int r = 0; r < n && rounds; r++)
            if (ag[r].size() != rounds) FAIL(ncclInvalidUsage, "clique %u: rank %d made %zu all-gather call(s) in this group, another rank %zu", q->id, r, ag[r].size(), rounds);
        for (size_t k = 0; k < rounds; k++) {
            const Op *first = ag[0][k];
            const size_t bytes = first->count * size_of(first->type);
            for (int r = 0; r < n; r++) {
                if (ag[r][k]->count != first->count || ag[r][k]->type != first->type)
                    FAIL(ncclInvalidArgument, "clique %u: all-gather count / type of rank %d (%zu, %d) differ from rank 0's (%zu, %d)", q->id, r, ag[r][k]->count, (int)ag[r][k]->type,
                         first->count, (int)first->type);
                for (int s = 0; s < n; s++) {
                    void *dst = (char *)ag[r][k]->recvbuff + (size_t)s * bytes;
                    if (s == r && dst == ag[s][k]->sendbuff) continue; // in place: rank r's own slot is already there
                    if (s != r && overlap(dst, bytes, ag[s][k]->sendbuff, bytes))
                        FAIL(ncclInvalidArgument, "clique %u: slot %d of rank %d's receive buffer overlaps rank %d's send buffer", q->id, s, r, s);
                    transfers.push_back({q->comms[s], q->comms[r], ag[s][k]->sendbuff, dst, bytes});
                }
            }
            n_allgathers++;
            n_ag_ranks += n;
            n_bytes += bytes * n * n;
        }
        // point to point
        uint64_t pairs_here = 0;
        for (auto &sv : sends) {
            auto rv = recvs.find(sv.first);
            const size_t have = rv == recvs.end() ? 0 : rv->second.size();
            if (have != sv.second.size())
                FAIL(ncclInvalidUsage, "clique %u: rank %d sends %zu message(s) to rank %d, which posts %zu receive(s) from it in this group", q->id, sv.first.first, sv.second.size(),
                     sv.first.second, have);
            for (size_t k = 0; k < have; k++) {
                const Op *s = sv.second[k], *r = rv->second[k];
                if (s->count != r->count || s->type != r->type)
                    FAIL(ncclInvalidArgument, "clique %u: message %zu of rank %d -> rank %d: sent (%zu, type %d), received as (%zu, type %d)", q->id, k, sv.first.first, sv.first.second,
                         s->count, (int)s->type, r->count, (int)r->type);
                const size_t bytes = s->count * size_of(s->type);
                if (overlap(s->sendbuff, bytes, r->recvbuff, bytes)) FAIL(ncclInvalidArgument, "clique %u: rank %d -> rank %d: source and destination overlap", q->id, sv.first.first, sv.first.second);
                transfers.push_back({s->comm, r->comm, s->sendbuff, r->recvbuff, bytes});
                pairs_here++;
                n_bytes += bytes;
            }
        }
        for (auto &rv : recvs)
            if (sends.find(rv.first) == sends.end())
                FAIL(ncclInvalidUsage, "clique %u: rank %d posts %zu receive(s) from rank %d, which sends nothing to it in this group", q->id, rv.first.second, rv.second.size(),
                     rv.first.first);
        if (pairs_here) {
            n_exchange_cliques++;
            n_pairs += pairs_here;
        }
    }
    // a destination must not be the source of another transfer of the same group: the order inside a group is undefined
    for (size_t a = 0; a < transfers.size(); a++)
        for (size_t b = 0; b < transfers.size(); b++)
            if (a != b && overlap(transfers[a].dst, transfers[a].bytes, transfers[b].src, transfers[b].bytes))
                FAIL(ncclInvalidArgument, "a destination of this group (%p, rank %d) overlaps a source of the same group (%p, rank %d): transfers of one group are unordered",
                     transfers[a].dst, transfers[a].to->rank, transfers[b].src, transfers[b].from->rank);
    // ---- perform.  Phase 1: every stream marks "work queued before the group is done"
    int caller_dev = 0;
    HIP_OK(hipGetDevice(&caller_dev));
    for (auto &kv : stream_of) {
        HIP_OK(hipSetDevice(kv.first->device));
        HIP_OK(hipEventRecord(kv.first->ready, kv.second));
    }
    // phase 2: every receiver waits for its senders and copies on its own stream
    std::map<FakeComm *, std::vector<FakeComm *>> readers_of; // sender -> receivers that read from it
    for (auto &kv : stream_of) {
        FakeComm *me = kv.first;
        HIP_OK(hipSetDevice(me->device));
        bool any = false;
        for (const Transfer &t : transfers)
            if (t.to == me && t.from != me) {
                HIP_OK(hipStreamWaitEvent(kv.second, t.from->ready, 0));
                readers_of[t.from].push_back(me);
                any = true;
            }
        for (const Transfer &t : transfers)
            if (t.to == me) {
                HIP_OK(hipMemcpyAsync(t.dst, t.src, t.bytes, hipMemcpyDeviceToDevice, kv.second));
                any = true;
            }
        if (any) HIP_OK(hipEventRecord(me->done, kv.second));
    }
    // phase 3: a sender's stream passes the group only when its readers have read
    for (auto &kv : readers_of) {
        HIP_OK(hipSetDevice(kv.first->device));
        for (FakeComm *reader : kv.second) HIP_OK(hipStreamWaitEvent(stream_of[kv.first], reader->done, 0));
    }
    HIP_OK(hipSetDevice(caller_dev));
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        g_cnt.groups++;
        g_cnt.allgathers += n_allgathers;
        g_cnt.allgather_ranks += n_ag_ranks;
        g_cnt.exchanges += n_exchange_cliques;
        g_cnt.matched_pairs += n_pairs;
        g_cnt.bytes += n_bytes;
    }
    logf("group ok: %zu call(s) on %zu communicator(s): %llu all-gather(s) over %llu rank-call(s), %llu matched send/recv pair(s), %llu byte(s)", ops.size(), stream_of.size(),
         (unsigned long long)n_allgathers, (unsigned long long)n_ag_ranks, (unsigned long long)n_pairs, (unsigned long long)n_bytes);
    return ncclSuccess;
}

ncclResult_t submit(const Op &op)
{
    if (t_depth > 0) {
        t_ops.push_back(op);
        return ncclSuccess;
    }
    if (op.comm->nranks > 1) {
        {
            std::lock_guard<std::mutex> lk(g_mutex);
            g_cnt.calls_outside_group++;
        }
        FAIL(ncclInvalidUsage, "rank %d of a %d-rank single-process clique called %s outside ncclGroupStart / ncclGroupEnd: real RCCL would block this thread for ever", op.comm->rank,
             op.comm->nranks, op.kind == ALLGATHER ? "ncclAllGather" : op.kind == SEND ? "ncclSend" : "ncclRecv");
    }
    std::vector<Op> one{op};
    return run_group(one);
}

} // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist)
{
    if (!comm || ndev <= 0) FAIL(ncclInvalidArgument, "ncclCommInitAll(%p, %d, ...)", (void *)comm, ndev);
    int count = 0, caller_dev = 0;
    HIP_OK(hipGetDeviceCount(&count));
    HIP_OK(hipGetDevice(&caller_dev));
    const bool shared = getenv("FAKE_RCCL_ALLOW_SHARED_DEVICE") != nullptr;
    for (int i = 0; i < ndev; i++) {
        const int dev = devlist ? devlist[i] : i;
        if (dev < 0 || dev >= count) FAIL(ncclInvalidArgument, "ncclCommInitAll: device %d of rank %d does not exist (%d device(s))", dev, i, count);
        for (int j = 0; j < i && !shared; j++)
            if ((devlist ? devlist[j] : j) == dev) FAIL(ncclInvalidUsage, "ncclCommInitAll: ranks %d and %d share device %d (real RCCL refuses duplicates)", j, i, dev);
    }
    Clique *q = new Clique();
    for (int i = 0; i < ndev; i++) {
        FakeComm *c = new FakeComm();
        c->rank = i;
        c->nranks = ndev;
        c->device = devlist ? devlist[i] : i;
        c->clique = q;
        HIP_OK(hipSetDevice(c->device));
        HIP_OK(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
        q->comms.push_back(c);
        comm[i] = reinterpret_cast<ncclComm_t>(c);
    }
    q->alive = ndev;
    HIP_OK(hipSetDevice(caller_dev));
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        q->id = g_next_clique++;
        g_cnt.cliques++;
        if ((uint64_t)ndev > g_cnt.largest_clique) g_cnt.largest_clique = ndev;
    }
    std::string devs;
    for (int i = 0; i < ndev; i++) devs += (i ? "," : "") + std::to_string(q->comms[i]->device);
    logf("clique %u: %d rank(s) on device(s) %s", q->id, ndev, devs.c_str());
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    FakeComm *c = comm_of(comm);
    if (!c) FAIL(ncclInvalidArgument, "ncclCommDestroy(%p): not a live communicator", (void *)comm);
    int caller_dev = 0;
    (void)hipGetDevice(&caller_dev);
    (void)hipSetDevice(c->device);
    (void)hipEventDestroy(c->ready);
    (void)hipEventDestroy(c->done);
    (void)hipSetDevice(caller_dev);
    c->magic = DEAD;
    Clique *q = c->clique;
    bool last;
    {
        std::lock_guard<std::mutex> lk(g_mutex);
        last = --q->alive == 0;
    }
    if (last) {
        logf("clique %u destroyed", q->id);
        for (FakeComm *x : q->comms) delete x;
        delete q;
    }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    if (t_depth++ == 0) t_ops.clear();
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) FAIL(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_group(ops);
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = comm_of(comm);
    if (!c) FAIL(ncclInvalidArgument, "ncclAllGather: %p is not a live communicator", (void *)comm);
    return submit(Op{ALLGATHER, c, sendbuff, recvbuff, sendcount, datatype, -1, stream});
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = comm_of(comm);
    if (!c) FAIL(ncclInvalidArgument, "ncclSend: %p is not a live communicator", (void *)comm);
    return submit(Op{SEND, c, sendbuff, nullptr, count, datatype, peer, stream});
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = comm_of(comm);
    if (!c) FAIL(ncclInvalidArgument, "ncclRecv: %p is not a live communicator", (void *)comm);
    return submit(Op{RECV, c, nullptr, recvbuff, count, datatype, peer, stream});
}

const char *ncclGetErrorString(ncclResult_t result)
{
    switch (result) {
    case ncclSuccess: return "no error (fake RCCL)";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake RCCL)";
    case ncclInvalidArgument: return "invalid argument (fake RCCL: see the FAIL line)";
    case ncclInvalidUsage: return "invalid usage (fake RCCL: see the FAIL line)";
    default: return "error (fake RCCL)";
    }
}

// test-side: is the interposer the one bound, and what has it seen
unsigned fake_rccl_present(void) { return 0xFA4E; }

void fake_rccl_counters(uint64_t out[10])
{
    std::lock_guard<std::mutex> lk(g_mutex);
    out[0] = g_cnt.cliques;
    out[1] = g_cnt.groups;
    out[2] = g_cnt.allgathers;
    out[3] = g_cnt.allgather_ranks;
    out[4] = g_cnt.exchanges;
    out[5] = g_cnt.matched_pairs;
    out[6] = g_cnt.bytes;
    out[7] = g_cnt.failures;
    out[8] = g_cnt.calls_outside_group;
    out[9] = g_cnt.largest_clique;
}

} // extern "C"
