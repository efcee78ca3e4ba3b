"""Experiment: aggregate MSM throughput with T host threads, each with its own stream and scratch arena, sharing one GPU."""
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


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    lib = ffi.load()
    n = 1 << k
    db = DeviceBuffer(n * 64)
    ffi.check(lib.panda_gen_bases(0, 1, 0, n, db.ptr, NULL_STREAM), "gen")
    for threads in (1, 2, 3):
        gms = [pgm.PandaGpuManager(0) for _ in range(threads)]
        ds = [DeviceBuffer(n * 32) for _ in range(threads)]
        dr = [DeviceBuffer(96) for _ in range(threads)]
        for i in range(threads):
            ffi.check(lib.panda_gen_scalars(0, 10 + i, 0, n, ds[i].ptr, NULL_STREAM), "gen")

        def work(i, count):
            cfg = ffi.MSMConfiguration(gms[i].mem_pool, gms[i].exec_stream.raw, db.ptr, ds[i].ptr, dr[i].ptr, k, 0)
            for _ in range(count):
                ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")

        ts = [threading.Thread(target=work, args=(i, 1)) for i in range(threads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(i, reps)) for i in range(threads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        dt = time.perf_counter() - t0
        print(f"2^{k} threads={threads}: {threads*reps*n/dt/1e6:8.1f} Mpts/s aggregate  ({dt/reps*1e3:.2f} ms per round of {threads})", flush=True)
        for d in ds + dr:
            d.free()
        for g in gms:
            g.deinit()
    db.free()


if __name__ == "__main__":
    main()
