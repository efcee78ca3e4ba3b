"""what the upload of the scalars still costs a from-host MSM (BN254 2^24, precomputed tables, PINNED host scalars),
by number of point ranges R: panda_msm_execute_from_host cuts the scalars into n/2^(R-1), n/2^(R-1), n/2^(R-2), ..., n/2 points (the sort
wants power-of-two ranges, so "a first range a quarter of the others" exists only inside this family: R = 3 is n/4, n/4, n/2), range r + 1
crossing PCIe beside the kernels of range r.  Prints per R: wall time of the call, and the same schedule on RESIDENT scalars (h_scalars = NULL:
what the ranges cost by themselves -- shorter lists, a merging fix-up per range).  usage: python tools/from_host_schedule.py [log_n=24]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log_n
lib = ffi.load()
db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)
ffi.check(lib.panda_gen_bases(0, 1, 0, n, db.ptr, NULL_STREAM), "gen")
ffi.check(lib.panda_gen_scalars(0, 2, 0, n, ds.ptr, NULL_STREAM), "gen")
ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, log_n, 0, NULL_STREAM), "pre")
host = C.c_void_p()
ffi.check(lib.panda_malloc_host(C.byref(host), n * 32), "malloc_host")
ffi.check(lib.panda_memcpy(host, ds.ptr, n * 32), "memcpy")
exec_s, h2d = ffi.PandaStream(), ffi.PandaStream()
ffi.check(lib.panda_stream_create(C.byref(exec_s), False), "stream")
ffi.check(lib.panda_stream_create(C.byref(h2d), False), "stream")
cfg = ffi.MSMConfiguration(ffi.PandaMemPool(), exec_s, db.ptr, ds.ptr, dr.ptr, log_n, 0)


def med(fn, reps=7):
    ts = []
    for i in range(reps + 2):
        t = time.perf_counter()
        fn()
        if i >= 2:
            ts.append(time.perf_counter() - t)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


resident = med(lambda: ffi.check(lib.panda_msm_execute_bn254(cfg), "msm"))
t = time.perf_counter()
for _ in range(5):
    ffi.check(lib.panda_memcpy(ds.ptr, host, n * 32), "memcpy")
copy_ms = (time.perf_counter() - t) / 5 * 1e3
print(f"BN254 2^{log_n}: resident scalars {resident:.3f} ms; one copy of the scalars from pinned memory {copy_ms:.3f} ms ({n * 32 / copy_ms / 1e6:.1f} GB/s); copy then call {resident + copy_ms:.3f} ms", flush=True)
for R in (1, 2, 3, 4, 5, 6, 7):
    a = med(lambda: ffi.check(lib.panda_msm_execute_from_host(0, cfg, host, R, h2d), "msm"))
    b = med(lambda: ffi.check(lib.panda_msm_execute_from_host(0, cfg, None, R, h2d), "msm"))
    first = n >> max(R - 1, 0)
    print(f"  R = {R}: from pinned host {a:7.3f} ms (exposed over resident: {a - resident:+.3f})   same ranges on resident scalars {b:7.3f} ms ({b - resident:+.3f})   "
          f"first range 2^{first.bit_length() - 1} points = {first * 32 / 2**20:.0f} MiB", flush=True)
