"""Experiment: aggregate MSM throughput with T host threads, each with its own stream and scratch arena, sharing
one GPU and one registered base set -- does the sort of one call (memory / LDS bound) run beside the k_accumulate of another (vector
issue bound)?  Optionally the streams are created with disjoint CU masks (hipExtStreamCreateWithCUMask).
usage: concurrent_msm.py <log_n> [reps] [tables 0|1] [cu_split: 0 = no masks, else CUs (of 256) given to thread 0; the others share the rest]"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402


def masked_stream(hip, first_cu, n_cu, total=256):
    words = (total + 31) // 32
    mask = (C.c_uint32 * words)()
    for cu in range(first_cu, first_cu + n_cu):
        mask[cu // 32] |= 1 << (cu % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return s


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    tables = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    split = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    lib = ffi.load()
    hip = C.CDLL("libamdhip64.so") if split else None
    n = 1 << k
    db = DeviceBuffer(n * 64)
    ffi.check(lib.panda_gen_bases(0, 1, 0, n, db.ptr, NULL_STREAM), "gen")
    if tables:
        ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, k, 0, NULL_STREAM), "precompute")
    else:
        ffi.check(lib.panda_msm_register_bases(0, db.ptr, k, NULL_STREAM), "register")
    for threads in (1, 2, 3):
        gms = [pgm.PandaGpuManager(0) for _ in range(threads)]
        ds = [DeviceBuffer(n * 32) for _ in range(threads)]
        dr = [DeviceBuffer(96) for _ in range(threads)]
        streams = [g.exec_stream.raw for g in gms]
        if split and threads > 1:
            rest = 256 - split
            streams = [ffi.PandaStream(masked_stream(hip, 0, split).value)] + [ffi.PandaStream(masked_stream(hip, split, rest).value) for _ in range(threads - 1)]
        for i in range(threads):
            ffi.check(lib.panda_gen_scalars(0, 10 + i, 0, n, ds[i].ptr, NULL_STREAM), "gen")

        def work(i, count):
            cfg = ffi.MSMConfiguration(gms[i].mem_pool, streams[i], db.ptr, ds[i].ptr, dr[i].ptr, k, 0)
            for _ in range(count):
                ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")

        ts = [threading.Thread(target=work, args=(i, 2)) for i in range(threads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(i, reps)) for i in range(threads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        dt = time.perf_counter() - t0
        how = f"CU masks {split}/{256 - split}" if split and threads > 1 else "no masks"
        print(f"2^{k} {'tables' if tables else 'plain '} threads={threads} ({how}): {threads*reps*n/dt/1e6:8.1f} Mpts/s aggregate  ({dt/reps*1e3:.3f} ms per round of {threads})", flush=True)
        for d in ds + dr:
            d.free()
        for g in gms:
            g.deinit()
    lib.panda_msm_unregister_bases(db.ptr)
    db.free()


if __name__ == "__main__":
    main()
