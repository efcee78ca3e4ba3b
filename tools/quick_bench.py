"""Quick per-phase timing of the MSM and NTT entry points on one GPU (development aid, not the contract bench)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402


def main():
    ks = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "16,20,22,24".split(","))]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = ffi.load()
    lib.panda_msm_set_phase_timing(2)
    gm = pgm.PandaGpuManager(0)
    names = [lib.panda_msm_phase_name(i).decode() for i in range(8)]
    for k in ks:
        n = 1 << k
        db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)
        t = time.time()
        ffi.check(lib.panda_gen_bases(0, 1, 0, n, db.ptr, NULL_STREAM), "gen")
        ffi.check(lib.panda_gen_scalars(0, 2, 0, n, ds.ptr, NULL_STREAM), "gen")
        tg = time.time() - t
        cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, 0)
        best = None
        for r in range(reps + 1):
            t = time.time()
            ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
            dt = time.time() - t
            ms = (C.c_float * 8)()
            lib.panda_msm_last_phase_ms(ms)
            if r and (best is None or dt < best[0]):
                best = (dt, list(ms))
        dt, ms = best
        print(f"MSM bn254 2^{k}: wall {dt*1e3:8.3f} ms  {n/dt/1e6:8.2f} Mpts/s  gen {tg:.2f}s | " + " ".join(f"{nm}={v:.3f}" for nm, v in zip(names, ms)), flush=True)
        for d in (db, ds, dr):
            d.free()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as po
    for k in [x for x in ks if x <= 26]:
        n = 1 << k
        om = po.root_of_unity(po.F_BN254_FR, k)
        da, dbb = DeviceBuffer(n * 32), DeviceBuffer(n * 32)
        ffi.check(lib.panda_gen_scalars(0, 3, 0, n, da.ptr, NULL_STREAM), "gen")
        flag = C.c_uint(0)
        cfg = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, da.ptr, dbb.ptr, C.c_void_p(om.ctypes.data), k, C.pointer(flag))
        best = 1e9
        for r in range(reps + 1):
            t = time.time()
            ffi.check(lib.panda_ntt_execute_bn254_v1(cfg), "ntt")
            dt = time.time() - t
            if r:
                best = min(best, dt)
        print(f"NTT bn254 2^{k}: wall {best*1e3:8.3f} ms  {n/best/1e9:8.3f} Gelem/s  {n*64/best/1e9:8.1f} GB/s algorithmic", flush=True)
        da.free()
        dbb.free()
    gm.deinit()


if __name__ == "__main__":
    main()
