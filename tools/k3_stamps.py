"""Phases of k3_merge's workgroups from in-kernel time stamps (tools/k3_stamps.sh builds the instrumented library first).
usage: k3_stamps.py [log_n = 24] [panda_msm_set_wide_merge mode = 0]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ffi.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libpanda-k3stamps.so")
    lib = ffi.load()
    gm = pgm.PandaGpuManager(0)
    n = 1 << k
    db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)
    ffi.check(lib.panda_gen_bases(0, 1, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(0, 2, 0, n, ds.ptr, NULL_STREAM), "gen")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, 0)
    ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, k, 0, gm.exec_stream.raw), "precompute")
    ffi.check(lib.panda_msm_set_wide_merge(mode), "mode")
    for _ in range(4):
        ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
    st = np.zeros(1024 * 8, dtype=np.uint64)
    raw = C.CDLL(ffi.LIB_PATH)
    raw.panda_debug_k3_stamps.argtypes = [C.c_void_p]
    assert raw.panda_debug_k3_stamps(st.ctypes.data) == 0
    st = st.reshape(1024, 8).astype(np.int64)
    seen = st[:, 0] != 0
    fast = seen & (st[:, 7] != 0)
    slow = seen & (st[:, 6] != 0)
    print(f"2^{k}, wide-merge mode {mode}: {int(seen.sum())} sampled workgroups (thread 0 of every 16th cell, the first 16384 cells), "
          f"{int(fast.sum())} on the read-once path, {int(slow.sum())} on the two-pass path")
    t0 = st[seen, 0].min()
    if fast.any():
        f = st[fast]
        names = [("setup: runs, prefix, two barriers", 0, 1), ("loads issued (with the run search)", 1, 2), ("wait for the loads", 2, 3), ("LDS atomics", 3, 4), ("barrier", 4, 5),
                 ("scan, offsets, ranking, copy out", 5, 7), ("whole workgroup", 0, 7)]
        print("  read-once path, microseconds (median / 10th / 90th percentile):")
        for nm, a, b in names:
            d = (f[:, b] - f[:, a]) / 100.0
            print(f"    {nm:36s} {np.median(d):7.2f} {np.percentile(d, 10):7.2f} {np.percentile(d, 90):7.2f}")
        print(f"    these workgroups start between {(f[:, 0].min() - t0) / 100.0:.1f} and {(f[:, 0].max() - t0) / 100.0:.1f} us of the kernel")
    if slow.any():
        f = st[slow]
        d = (f[:, 6] - f[:, 0]) / 100.0
        print(f"  two-pass path: whole workgroup {np.median(d):.2f} us (10th {np.percentile(d, 10):.2f}, 90th {np.percentile(d, 90):.2f}); "
              f"they start between {(f[:, 0].min() - t0) / 100.0:.1f} and {(f[:, 0].max() - t0) / 100.0:.1f} us of the kernel")
    last = max(st[:, 6].max(), st[:, 7].max())
    print(f"  span of the sampled workgroups: {(last - t0) / 100.0:.1f} us")
    lib.panda_msm_set_wide_merge(0)
    lib.panda_msm_unregister_bases(db.ptr)


if __name__ == "__main__":
    main()
