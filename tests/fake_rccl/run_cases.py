"""Child process of tests/test_fake_rccl.py (never collected by pytest itself): runs the sharded C entry points over the
PANDA_MULTI_RCCL transport with 2 / 4 / 8 ranks on device 0, with tests/fake_rccl/libfake_rccl.so preloaded in place of RCCL.

It must be started as a FRESH process with
    LD_PRELOAD=<repo>/tests/fake_rccl/libfake_rccl.so   PANDA_TEST_SHARED_DEVICE_RCCL=1   FAKE_RCCL_ALLOW_SHARED_DEVICE=1
(the parent sets them in the child's environment; nothing is re-executed after the GPU was touched).

Every case is the body of a `-m gpu` parity test of tests/test_gpu_parity.py called with the RCCL transport instead of the loopback
one, so the answers are checked against the oracle exactly as there.  After each case the interposer's counters must show that every
exchange matched ranks x ranks send/receive pairs, that every all-gather was entered by all ranks, and that no validation failed.
The second half turns the interposer on itself: calls that real RCCL would reject or hang on must be refused (so a green product run
means something), and the copies must respect stream order.  Prints one JSON line `FAKE_RCCL_RESULT {...}` at the end.
"""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

FAKE = os.path.join(HERE, "libfake_rccl.so")
NAMES = ["cliques", "groups", "allgathers", "allgather_ranks", "exchanges", "matched_pairs", "bytes", "failures", "calls_outside_group", "largest_clique"]


def counters(fake):
    out = (C.c_uint64 * 10)()
    fake.fake_rccl_counters(out)
    return dict(zip(NAMES, [int(v) for v in out]))


def main():
    assert FAKE in os.environ.get("LD_PRELOAD", ""), "start me with LD_PRELOAD=libfake_rccl.so"
    assert os.environ.get("PANDA_TEST_SHARED_DEVICE_RCCL") and os.environ.get("FAKE_RCCL_ALLOW_SHARED_DEVICE")
    fake = C.CDLL(FAKE)
    fake.fake_rccl_present.restype = C.c_uint
    assert fake.fake_rccl_present() == 0xFA4E

    import test_gpu_parity as T
    from panda_amd import gpu_ffi as ffi
    from panda_amd import gpu_manager as pgm

    lib = ffi.load()
    # the product library's RCCL references must be bound to the interposer, not to librccl
    # (a lookup in the global scope, which is what the library's PLT uses; the per-case counters below are the real proof)
    assert C.cast(C.CDLL(None).ncclAllGather, C.c_void_p).value == C.cast(fake.ncclAllGather, C.c_void_p).value, "ncclAllGather is not bound to the interposer"

    gm = pgm.PandaGpuManager(0)
    R = ffi.MULTI_RCCL
    cases = []
    for ranks in (2, 4, 8):
        cases.append((f"msm[{ranks}]", ranks, lambda ranks=ranks: T.test_c_abi_multi_gpu_msm(gm, ranks, R)))
    cases += [
        ("msm_from_host[2,pinned,tables]", 2, lambda: T.test_c_abi_multi_gpu_msm_from_host(gm, 2, R, True, True)),
        ("msm_from_host[8,pageable,tables]", 8, lambda: T.test_c_abi_multi_gpu_msm_from_host(gm, 8, R, False, True)),
        ("msm_from_host[4,pinned,plain]", 4, lambda: T.test_c_abi_multi_gpu_msm_from_host(gm, 4, R, True, False)),
        ("ntt[2,2^9]", 2, lambda: T.test_c_abi_multi_gpu_ntt(gm, 2, R, 9)),
        ("ntt[4,2^14]", 4, lambda: T.test_c_abi_multi_gpu_ntt(gm, 4, R, 14)),
        ("ntt[8,2^21]", 8, lambda: T.test_c_abi_multi_gpu_ntt(gm, 8, R, 21)),
        ("ntt_batch[2,2^10,4]", 2, lambda: T.test_c_abi_multi_gpu_ntt_batch(gm, 2, R, 10, 4)),
        ("ntt_batch[4,2^14,3]", 4, lambda: T.test_c_abi_multi_gpu_ntt_batch(gm, 4, R, 14, 3)),
        ("ntt_batch[8,2^20,5]", 8, lambda: T.test_c_abi_multi_gpu_ntt_batch(gm, 8, R, 20, 5)),
        ("msm_other_curves[4]", 4, lambda: T.test_c_abi_multi_gpu_msm_other_curves(gm, 4, R)),
    ]
    for what in ("msm", "msm_from_host", "ntt", "ntt_batch", "ntt_bls12_377", "ntt_bls12_381"):
        for ranks in (2, 8):
            cases.append((f"sharded_over[{what},{ranks}]", ranks, lambda what=what, ranks=ranks: T._sharded_over([0] * ranks, R, what)))
    # two handles driven from two host threads at once (thread-local RCCL groups must not mix)
    orig = T.multi_gpu.MultiGpu

    def two_threads():
        class RcclMultiGpu(orig):
            def __init__(self, devices, transport=R):
                super().__init__(devices, R)

        T.multi_gpu.MultiGpu = RcclMultiGpu
        try:
            T.test_c_abi_multi_gpu_handles_from_two_threads(gm)
        finally:
            T.multi_gpu.MultiGpu = orig

    cases.append(("two_handles_two_threads[2]", 2, two_threads))

    report = []
    before = counters(fake)
    for name, ranks, fn in cases:
        fn()
        now = counters(fake)
        d = {k: now[k] - before[k] for k in NAMES}
        before = now
        assert d["failures"] == 0 and d["calls_outside_group"] == 0, (name, d)
        assert d["groups"] > 0, (name, "no RCCL group was closed: the RCCL transport did not run", d)
        assert d["matched_pairs"] == d["exchanges"] * ranks * ranks, (name, d)
        assert d["allgather_ranks"] == d["allgathers"] * ranks, (name, d)
        assert d["groups"] == d["allgathers"] + d["exchanges"], (name, d)
        report.append({"case": name, "ranks": ranks, **{k: d[k] for k in ("groups", "allgathers", "exchanges", "matched_pairs", "bytes")}})
        print(f"ok {name}: {d['groups']} group(s), {d['allgathers']} all-gather(s) x {ranks} ranks, {d['exchanges']} exchange(s) x {ranks * ranks} pairs", flush=True)
    gm.deinit()
    product = counters(fake)
    assert product["failures"] == 0 and product["largest_clique"] == 8

    # ---------------------------------------------------------------- the interposer against itself
    from gpu_util import DeviceBuffer

    OK, ARG, USAGE = 0, 4, 5
    for f in (fake.ncclAllGather, fake.ncclSend, fake.ncclRecv, fake.ncclCommInitAll, fake.ncclCommDestroy, fake.ncclGroupStart, fake.ncclGroupEnd):
        f.restype = C.c_int
    fake.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    fake.ncclSend.argtypes = fake.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fake.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    fake.ncclCommDestroy.argtypes = [C.c_void_p]
    comms = (C.c_void_p * 2)()
    assert fake.ncclCommInitAll(comms, 2, (C.c_int * 2)(0, 0)) == OK
    streams = [ffi.PandaStream(), ffi.PandaStream()]
    for s in streams:
        ffi.check(lib.panda_stream_create(C.byref(s), False), "stream")
    st = [s.handle for s in streams]
    n = 1 << 20
    a, b = DeviceBuffer(2 * n), DeviceBuffer(2 * n)
    host = np.zeros(n, dtype=np.uint8)
    refused = []

    def expect(code, what, fn):
        got = fn()
        assert got == code, (what, got, code)
        refused.append(what)

    def group(*calls):
        assert fake.ncclGroupStart() == OK
        for c in calls:
            r = c()
            assert r == OK, r  # queued: errors surface at ncclGroupEnd
        return fake.ncclGroupEnd()

    A, B = a.ptr.value, b.ptr.value
    expect(USAGE, "all-gather outside a group on a 2-rank clique", lambda: fake.ncclAllGather(A, A, n, 0, comms[0], st[0]))
    expect(USAGE, "send outside a group on a 2-rank clique", lambda: fake.ncclSend(A, n, 0, 1, comms[0], st[0]))
    expect(USAGE, "only one rank enters the all-gather", lambda: group(lambda: fake.ncclAllGather(A, A, n, 0, comms[0], st[0])))
    expect(ARG, "in-place all-gather from the wrong slot", lambda: group(lambda: fake.ncclAllGather(A, A, n, 0, comms[0], st[0]),      # rank 0 in place: fine
                                                                       lambda: fake.ncclAllGather(B, B, n, 0, comms[1], st[1])))     # rank 1 must send from B + n
    expect(ARG, "all-gather counts differ", lambda: group(lambda: fake.ncclAllGather(A, A, n, 0, comms[0], st[0]), lambda: fake.ncclAllGather(B + n // 2, B, n // 2, 0, comms[1], st[1])))
    expect(USAGE, "send without a receive", lambda: group(lambda: fake.ncclSend(A, n, 0, 1, comms[0], st[0])))
    expect(USAGE, "receive without a send", lambda: group(lambda: fake.ncclRecv(B, n, 0, 0, comms[1], st[1])))
    expect(USAGE, "receive posted for the wrong peer", lambda: group(lambda: fake.ncclSend(A, n, 0, 1, comms[0], st[0]), lambda: fake.ncclRecv(B, n, 0, 1, comms[1], st[1])))
    expect(ARG, "send and receive counts differ", lambda: group(lambda: fake.ncclSend(A, n, 0, 1, comms[0], st[0]), lambda: fake.ncclRecv(B, n // 2, 0, 0, comms[1], st[1])))
    expect(ARG, "a transfer that runs past its allocation", lambda: group(lambda: fake.ncclSend(A, 2 * n, 0, 1, comms[0], st[0]), lambda: fake.ncclRecv(B, 64 * n, 0, 0, comms[1], st[1])))
    expect(ARG, "host memory as a send buffer", lambda: group(lambda: fake.ncclSend(host.ctypes.data, n, 0, 1, comms[0], st[0]), lambda: fake.ncclRecv(B, n, 0, 0, comms[1], st[1])))
    expect(ARG, "peer outside the clique", lambda: group(lambda: fake.ncclSend(A, n, 0, 2, comms[0], st[0])))
    expect(USAGE, "two streams for one communicator in a group", lambda: group(lambda: fake.ncclSend(A, n, 0, 1, comms[0], st[0]), lambda: fake.ncclSend(A + n, n, 0, 1, comms[0], st[1]),
                                                                                 lambda: fake.ncclRecv(B, n, 0, 0, comms[1], st[1]), lambda: fake.ncclRecv(B + n, n, 0, 0, comms[1], st[1])))
    expect(ARG, "a destination that is another transfer's source", lambda: group(lambda: fake.ncclSend(A, n, 0, 1, comms[0], st[0]), lambda: fake.ncclRecv(B, n, 0, 0, comms[1], st[1]),
                                                                                    lambda: fake.ncclSend(B, n, 0, 0, comms[1], st[1]), lambda: fake.ncclRecv(A + n, n, 0, 1, comms[0], st[0])))
    assert fake.ncclGroupEnd() == USAGE  # no group open
    refused.append("ncclGroupEnd without ncclGroupStart")
    after_negative = counters(fake)
    assert after_negative["failures"] - product["failures"] == len(refused), (after_negative, len(refused))

    # positive: in-place all-gather and a 2 x 2 exchange move the right bytes, and wait for the work queued BEFORE them on the sender's stream
    big = 1 << 30
    src, dst = DeviceBuffer(big), DeviceBuffer(big)
    ffi.check(lib.panda_memset(src.ptr, 0x11, big), "memset")
    ffi.check(lib.panda_memset(dst.ptr, 0x22, big), "memset")
    ffi.check(lib.panda_memset_async(src.ptr, 0x5A, big, streams[0]), "memset_async")  # still running when the group is queued
    assert group(lambda: fake.ncclSend(src.ptr.value, big, 0, 1, comms[0], st[0]), lambda: fake.ncclRecv(dst.ptr.value, big, 0, 0, comms[1], st[1])) == OK
    ffi.check(lib.panda_memset_async(src.ptr, 0x77, big, streams[0]), "memset_async")  # must not overtake the receiver's read
    ffi.check(lib.panda_stream_sync(streams[1]), "sync")
    got = dst.to_host(np.uint8)
    assert (got == 0x5A).all(), "the receiver's copy did not respect the sender's stream order"
    ffi.check(lib.panda_stream_sync(streams[0]), "sync")
    for buf, fill in ((a, 1), (b, 2)):
        ffi.check(lib.panda_memset(buf.ptr, 0, 2 * n), "memset")
        ffi.check(lib.panda_memset(C.c_void_p(buf.ptr.value + (fill - 1) * n), fill, n), "memset")
    assert group(lambda: fake.ncclAllGather(A, A, n, 0, comms[0], st[0]), lambda: fake.ncclAllGather(B + n, B, n, 0, comms[1], st[1])) == OK
    for s in streams:
        ffi.check(lib.panda_stream_sync(s), "sync")
    want = np.concatenate([np.full(n, 1, np.uint8), np.full(n, 2, np.uint8)])
    assert (a.to_host(np.uint8) == want).all() and (b.to_host(np.uint8) == want).all()
    for comm in comms:
        assert fake.ncclCommDestroy(comm) == OK
    assert fake.ncclCommDestroy(comms[0]) == ARG  # already destroyed
    for d in (a, b, src, dst):
        d.free()
    for s in streams:
        lib.panda_stream_destroy(s)
    print("FAKE_RCCL_RESULT " + json.dumps({"cases": report, "product_totals": product, "refused": refused}), flush=True)


if __name__ == "__main__":
    main()
